// b9dist.cpp -- see b9dist.hpp.  Host-only code: HIP runtime API + RCCL; compiled with hipcc for their headers.
#include "b9dist.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

namespace b9h {

// One string per launch, shared by its ranks: the launcher's own (B9_LAUNCH_NONCE), or torchrun's run id + restart count.
// Empty when the environment names no launch (srun / mpirun exporting only RANK): see RcclExchange.
std::string launch_nonce()
{
    std::string s;
    if (const char *n = std::getenv("B9_LAUNCH_NONCE")) s = n;
    else if (const char *run = std::getenv("TORCHELASTIC_RUN_ID")) {
        const char *restart = std::getenv("TORCHELASTIC_RESTART_COUNT");
        s = std::string(run) + "_" + (restart ? restart : "0");
    }
    for (char &c : s) if (!(std::isalnum((unsigned char)c) || c == '_' || c == '-')) c = '_';
    return s;
}

// Test hook (tests/test_launcher.py): B9_TEST_STALL="<where>:<rank>" parks that rank for ever at the named point, so that
// the launchers' start-up deadline can be exercised without a GPU or a real RCCL hang.
void test_stall(const char *where, int rank)
{
    const char *v = std::getenv("B9_TEST_STALL");
    if (!v) return;
    if (std::string(v) == std::string(where) + ":" + std::to_string(rank))
        for (;;) std::this_thread::sleep_for(std::chrono::seconds(3600));
}

// What a multi-rank run must be able to say about itself: the communicator has exactly `world` ranks and they sit on
// `world` DISTINCT GPUs (PCI bus ids gathered through the communicator).  Empty string = fine.
std::string group_error(int world, int comm_ranks, const std::string &devices_csv)
{
    if (world <= 1) return "";
    if (comm_ranks != world)
        return "the RCCL communicator reports " + std::to_string(comm_ranks) + " rank(s), the launch has " + std::to_string(world);
    std::vector<std::string> ids;
    size_t a = 0;
    while (a <= devices_csv.size()) {
        const size_t b = devices_csv.find(',', a);
        const std::string id = devices_csv.substr(a, b == std::string::npos ? std::string::npos : b - a);
        if (!id.empty()) ids.push_back(id);
        if (b == std::string::npos) break;
        a = b + 1;
    }
    if ((int)ids.size() != world)
        return "the communicator gathered " + std::to_string(ids.size()) + " device id(s) for " + std::to_string(world) + " rank(s): " + devices_csv;
    for (size_t i = 0; i < ids.size(); ++i)
        for (size_t j = i + 1; j < ids.size(); ++j)
            if (ids[i] == ids[j])
                return "ranks " + std::to_string(i) + " and " + std::to_string(j) + " share the GPU " + ids[i] +
                       ": a --gpus " + std::to_string(world) + " run needs " + std::to_string(world) + " distinct devices (" + devices_csv + ")";
    return "";
}

namespace {

[[noreturn]] void fail(const std::string &msg) { throw std::runtime_error("b9dist: " + msg); }

#define HIPX(call)                                                                              \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) fail(std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)
#define NCCLX(call)                                                                             \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) fail(std::string(#call) + ": " + ncclGetErrorString(r_));        \
    } while (0)

class LocalExchange final : public Exchange {
  public:
    int rank() const override { return 0; }
    int world() const override { return 1; }
    void start_host(int slot, const double *rows, size_t count) override { buf[slot & 1].assign(rows, rows + count); }
    const double *finish(int slot) override { return buf[slot & 1].data(); }
    const char *name() const override { return "none (one rank)"; }

  private:
    std::vector<double> buf[2];
};

class RcclExchange final : public Exchange {
  public:
    RcclExchange(int rank, int world, const std::string &dir, int device, double timeout_s) : rank_(rank), world_(world), device_(device)
    {
        if (world < 1 || rank < 0 || rank >= world) fail("bad rank / world");
        // The id file carries this LAUNCH's nonce in its name, so an id a crashed or restarted attempt left in a reused
        // directory (torchrun elastic restart, a user-supplied B9_DIST_DIR) is never read; the directory must be ours.
        // A launcher that names no launch (only RANK / WORLD_SIZE exported) would make every attempt share one file name,
        // and a rank > 0 could pick up a dead attempt's id and wait in ncclCommInitRank for ever: refused.
        const std::string nonce = launch_nonce();
        if (nonce.empty() && world > 1)
            fail("no launch id: export B9_LAUNCH_NONCE=<a string unique to this launch, the same on every rank> "
                 "(our own launchers and torchrun provide one)");
        test_stall("before-init", rank);
        HIPX(hipSetDevice(device));
        ncclUniqueId id;
        const std::string path = dir + "/rccl_id." + (nonce.empty() ? std::string("single") : nonce);
        if (rank == 0) {
            NCCLX(ncclGetUniqueId(&id));
            if (::mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) fail("cannot create " + dir);
        }
        {   // (ranks > 0 check it once it exists: see the wait below)
            struct stat sb;
            if (::stat(dir.c_str(), &sb) == 0 && (sb.st_uid != ::geteuid() || (sb.st_mode & 022)))
                fail(dir + " is not owned by this user or is writable by others: refusing to bootstrap RCCL through it");
        }
        if (rank == 0) {
            const std::string tmp = path + ".tmp";
            std::remove(path.c_str());
            std::remove(tmp.c_str());
            FILE *f = std::fopen(tmp.c_str(), "wb");
            if (!f) fail("cannot write " + tmp);
            const size_t n = std::fwrite(&id, 1, sizeof id, f);
            std::fclose(f);
            if (n != sizeof id || std::rename(tmp.c_str(), path.c_str()) != 0) fail("cannot publish " + path);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                FILE *f = std::fopen(path.c_str(), "rb");
                if (f) {
                    const size_t n = std::fread(&id, 1, sizeof id, f);
                    std::fclose(f);
                    if (n == sizeof id) break;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                    fail("rank " + std::to_string(rank) + " timed out waiting for " + path);
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
            struct stat sb;
            if (::stat(dir.c_str(), &sb) != 0 || sb.st_uid != ::geteuid() || (sb.st_mode & 022))
                fail(dir + " is not owned by this user or is writable by others: refusing to bootstrap RCCL through it");
        }
        // (ncclCommInitRank has no deadline of its own: a launcher that made B9_DIST_DIR watches for the ready markers
        //  written below and ends the launch when they do not appear in time -- cli_common.cpp, bench.py)
        NCCLX(ncclCommInitRank(&comm_, world, id, rank));
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            HIPX(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, greatest));
        else
            HIPX(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        for (auto &s : slot_) HIPX(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIPX(hipMalloc((void **)&d_scalar_, sizeof(double) * 2));
        HIPX(hipHostMalloc((void **)&h_scalar_, sizeof(double) * 2, hipHostMallocDefault));
        barrier();                                            // everyone has read the id
        if (rank == 0) {
            std::remove(path.c_str());
            // The directory is removed here only when it is the DEFAULT one this library made up (nobody else would remove it,
            // and it is empty now).  A directory a launcher of ours made is the launcher's to remove (B9_LAUNCHER_OWNS_DIR), and
            // one the user named through B9_DIST_DIR is the user's: never removed.
            if (!std::getenv("B9_LAUNCHER_OWNS_DIR") && !std::getenv("B9_DIST_DIR")) (void)::rmdir(dir.c_str());
        }
        // what the communicator itself says about the group: its rank count, and every rank's GPU (host name + PCI bus id:
        // identical nodes have identical bus ids), all-gathered through it -- a record that N distinct devices took part, not
        // an echo of the arguments
        NCCLX(ncclCommCount(comm_, &comm_count_));
        {
            constexpr int L = 96;
            char mine[L] = {0};
            char host[48] = {0}, bus[32] = {0};
            if (::gethostname(host, sizeof host - 1) != 0) std::snprintf(host, sizeof host, "host");
            HIPX(hipDeviceGetPCIBusId(bus, sizeof bus, device));
            for (char *c = host; *c; ++c) if (*c == ',') *c = '_';       // (the ids travel as a comma-separated list)
            std::snprintf(mine, L, "%s/%s", host, bus);
            char *d_all = nullptr;
            HIPX(hipMalloc((void **)&d_all, (size_t)L * world));
            HIPX(hipMemcpyAsync(d_all + (size_t)L * rank, mine, L, hipMemcpyHostToDevice, stream_));
            NCCLX(ncclAllGather(d_all + (size_t)L * rank, d_all, L, ncclChar, comm_, stream_));
            std::vector<char> all((size_t)L * world);
            HIPX(hipMemcpyAsync(all.data(), d_all, all.size(), hipMemcpyDeviceToHost, stream_));
            HIPX(hipStreamSynchronize(stream_));
            (void)hipFree(d_all);
            for (int r = 0; r < world; ++r) {
                all[(size_t)L * r + L - 1] = '\0';
                devices_ += (r ? "," : "") + std::string(&all[(size_t)L * r]);
            }
        }
        {
            const std::string bad = group_error(world, comm_count_, devices_);
            if (!bad.empty()) fail(bad);
        }
        test_stall("after-init", rank);
        if (std::getenv("B9_LAUNCHER_OWNS_DIR")) {   // ready marker for OUR launchers' start-up deadline (they remove it; nobody else would:
                                                     // a srun / mpirun user who exports B9_LAUNCH_NONCE by hand gets no markers)
            const std::string ready = dir + "/ready." + nonce + "." + std::to_string(rank);
            if (FILE *f = std::fopen(ready.c_str(), "w")) std::fclose(f);
        }
    }

    ~RcclExchange() override
    {
        (void)hipSetDevice(device_);
        (void)hipStreamSynchronize(stream_);
        if (comm_) (void)ncclCommDestroy(comm_);
        for (auto &s : slot_) {
            if (s.d_send) (void)hipFree(s.d_send);
            if (s.d_recv) (void)hipFree(s.d_recv);
            if (s.h_send) (void)hipHostFree(s.h_send);
            if (s.h_recv) (void)hipHostFree(s.h_recv);
            if (s.done) (void)hipEventDestroy(s.done);
        }
        if (d_scalar_) (void)hipFree(d_scalar_);
        if (h_scalar_) (void)hipHostFree(h_scalar_);
        if (stream_) (void)hipStreamDestroy(stream_);
    }

    int rank() const override { return rank_; }
    int world() const override { return world_; }
    const char *name() const override { return "RCCL all-gather (ncclAllGather over xGMI, device buffers, own high-priority stream)"; }

    int comm_ranks() const override { return comm_count_; }
    std::string devices() const override { return devices_; }

    bool reads_device_rows() const override { return true; }
    bool start_device(int slot, const double *d_rows, void *ready_event, size_t count) override
    {
        Slot &s = slot_[slot & 1];
        ensure(s, count);
        HIPX(hipSetDevice(device_));
        if (ready_event) HIPX(hipStreamWaitEvent(stream_, static_cast<hipEvent_t>(ready_event), 0));
        gather(s, d_rows, count);
        return true;
    }

    void start_host(int slot, const double *rows, size_t count) override
    {
        Slot &s = slot_[slot & 1];
        ensure(s, count);
        HIPX(hipSetDevice(device_));
        std::memcpy(s.h_send, rows, count * sizeof(double));
        HIPX(hipMemcpyAsync(s.d_send, s.h_send, count * sizeof(double), hipMemcpyHostToDevice, stream_));
        gather(s, s.d_send, count);
    }

    const double *finish(int slot) override
    {
        Slot &s = slot_[slot & 1];
        HIPX(hipEventSynchronize(s.done));
        return s.h_recv;
    }

    double all_reduce_max(double v) override
    {
        HIPX(hipSetDevice(device_));
        h_scalar_[0] = v;
        HIPX(hipMemcpyAsync(d_scalar_, h_scalar_, sizeof(double), hipMemcpyHostToDevice, stream_));
        NCCLX(ncclAllReduce(d_scalar_, d_scalar_ + 1, 1, ncclDouble, ncclMax, comm_, stream_));
        HIPX(hipMemcpyAsync(h_scalar_ + 1, d_scalar_ + 1, sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIPX(hipStreamSynchronize(stream_));
        return h_scalar_[1];
    }

    void barrier() override { (void)all_reduce_max(0.0); }

  private:
    struct Slot {
        double *d_send = nullptr, *d_recv = nullptr, *h_send = nullptr, *h_recv = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
    };

    void ensure(Slot &s, size_t count)
    {
        if (count <= s.cap) return;
        HIPX(hipSetDevice(device_));
        HIPX(hipStreamSynchronize(stream_));
        if (s.d_send) (void)hipFree(s.d_send);
        if (s.d_recv) (void)hipFree(s.d_recv);
        if (s.h_send) (void)hipHostFree(s.h_send);
        if (s.h_recv) (void)hipHostFree(s.h_recv);
        s.d_send = s.d_recv = s.h_send = s.h_recv = nullptr;
        HIPX(hipMalloc((void **)&s.d_send, count * sizeof(double)));
        HIPX(hipMalloc((void **)&s.d_recv, count * world_ * sizeof(double)));
        HIPX(hipHostMalloc((void **)&s.h_send, count * sizeof(double), hipHostMallocDefault));
        HIPX(hipHostMalloc((void **)&s.h_recv, count * world_ * sizeof(double), hipHostMallocDefault));
        s.cap = count;
    }

    void gather(Slot &s, const double *d_src, size_t count)
    {
        NCCLX(ncclAllGather(d_src, s.d_recv, count, ncclDouble, comm_, stream_));
        HIPX(hipMemcpyAsync(s.h_recv, s.d_recv, count * world_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIPX(hipEventRecord(s.done, stream_));
    }

    int rank_, world_, device_, comm_count_ = 0;
    std::string devices_;
    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    Slot slot_[2];
    double *d_scalar_ = nullptr, *h_scalar_ = nullptr;
};

}  // namespace

std::unique_ptr<Exchange> make_local_exchange() { return std::unique_ptr<Exchange>(new LocalExchange()); }

std::unique_ptr<Exchange> make_rccl_exchange(int rank, int world, const std::string &dir, int device, double timeout_s)
{
    return std::unique_ptr<Exchange>(new RcclExchange(rank, world, dir, device, timeout_s));
}

std::string default_bootstrap_dir()
{
    if (const char *d = std::getenv("B9_DIST_DIR")) return d;
    const char *port = std::getenv("MASTER_PORT");
    const char *tmp = std::getenv("TMPDIR");
    return std::string(tmp && *tmp ? tmp : "/tmp") + "/b9dist_" + std::to_string((long)getppid()) + "_" + (port ? port : "0");
}

bool forced_ranks()
{
    const char *v = std::getenv("B9_FORCE_RANKS");
    return v && std::atoi(v) != 0;
}

void rank_from_env(int &rank, int &world, int &local_rank)
{
    auto get = [](const char *a, const char *b, int def) {
        const char *v = std::getenv(a);
        if (!v) v = std::getenv(b);
        return v ? std::atoi(v) : def;
    };
    rank = get("B9_RANK", "RANK", 0);
    world = get("B9_WORLD_SIZE", "WORLD_SIZE", 1);
    local_rank = get("B9_LOCAL_RANK", "LOCAL_RANK", rank);
}

}  // namespace b9h
