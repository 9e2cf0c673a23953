// b9dist.hpp -- one rank of a walker-parallel run: an RCCL communicator over the GPUs of one node (xGMI) and the
// all-gather of per-walker block summary rows the adaptive proposal step needs (BASELINE.json north_star: "partition
// independent walkers/chains across the 8 GPUs of one node with an RCCL all-gather of log-posteriors over xGMI for
// the adaptive proposal step").  No torch, no MPI: ranks are processes started by the launcher (singlePopMcmc --gpus N,
// bench.py --gpus N, or torchrun), and the 128-byte RCCL unique id travels through a file in a directory they share.
//
// There is no reference counterpart: the reference runs one chain on CPU threads [RECALL].
#pragma once
#include <cstddef>
#include <memory>
#include <string>

namespace b9h {

// How summary rows travel between ranks.  `slot` (0 / 1) names one of two exchanges that may be in flight at once.
class Exchange {
  public:
    virtual ~Exchange() = default;
    virtual int rank() const = 0;
    virtual int world() const = 0;
    // Rows already in HBM (b9_mcmc_block::d_rows): the collective is enqueued on the exchange's own stream behind
    // `ready_event` (a hipEvent_t) and reads them in place; nothing waits on the host.  false = not supported by this
    // exchange (the caller then passes host rows to start_host).
    virtual bool start_device(int slot, const double *d_rows, void *ready_event, size_t count) { (void)slot; (void)d_rows; (void)ready_event; (void)count; return false; }
    virtual bool reads_device_rows() const { return false; }      // true: start_device works (the runner then records the rows' event)
    // Rows on the host.
    virtual void start_host(int slot, const double *rows, size_t count) = 0;
    // Blocks until the exchange of `slot` is complete; returns world() * count doubles in rank order (valid until the
    // slot is started again).
    virtual const double *finish(int slot) = 0;
    // max over ranks (timing) and a barrier; defaults are the one-rank answers
    virtual double all_reduce_max(double v) { return v; }
    virtual void barrier() {}
    virtual const char *name() const = 0;
    // What the communicator reports about itself: its rank count (ncclCommCount; 0 = no communicator) and the PCI bus
    // ids of the ranks' GPUs in rank order, comma-separated, gathered THROUGH the communicator.
    virtual int comm_ranks() const { return 0; }
    virtual std::string devices() const { return ""; }
};

// One rank: every "exchange" is a copy.
std::unique_ptr<Exchange> make_local_exchange();

// RCCL over xGMI.  Collective among `world` processes, one per GPU: rank 0 creates the unique id and publishes it as
// <dir>/rccl_id.<nonce> (written under a temporary name, then renamed; the nonce names the launch -- B9_LAUNCH_NONCE, or
// torchrun's run id and restart count -- so an id left behind by an earlier attempt is never picked up; <dir> must
// belong to this user and not be writable by others); the others wait for the file (up to timeout_s), and every rank
// calls ncclCommInitRank on `device`, then -- under OUR launchers (B9_LAUNCH_NONCE set), which watch for them and remove
// them -- touches <dir>/ready.<nonce>.<rank>.  With world > 1 and no launch id in the environment (a launcher exporting
// only RANK) the constructor refuses before any GPU call and asks for B9_LAUNCH_NONCE.  The exchange owns a non-blocking HIP stream of the HIGHEST priority
// (the sampler's context stream has the lowest: the gather gets in between two of its kernels), device and pinned
// host buffers for two slots, and reads device rows in place.  Throws std::runtime_error.
std::unique_ptr<Exchange> make_rccl_exchange(int rank, int world, const std::string &dir, int device, double timeout_s = 120.0);

// The directory the ranks of THIS launch share for the id: $B9_DIST_DIR when the launcher made one, else a name derived
// from the parent process and MASTER_PORT (torchrun workers share both).
std::string default_bootstrap_dir();

// RANK / WORLD_SIZE / LOCAL_RANK of this process as the launchers export them (B9_RANK ... take precedence); 0 / 1 / 0
// when absent.
void rank_from_env(int &rank, int &world, int &local_rank);

// true when this process is a rank of one of OUR launchers (singlePopMcmc --gpus N, bench.py --gpus N) and was told to use
// the multi-rank route whatever the rank count (B9_FORCE_RANKS=1: --forceRanks / --force-ranks with one GPU)
bool forced_ranks();

// One string per launch, shared by its ranks (names the RCCL id file): B9_LAUNCH_NONCE, or torchrun's run id + restart
// count; empty when the environment names no launch.
std::string launch_nonce();

// "" when a communicator of `comm_ranks` ranks on the GPUs `devices_csv` (PCI bus ids, comma-separated, in rank order) is
// what a launch of `world` ranks must have: that many ranks on that many DISTINCT devices; else what is wrong.  The RCCL
// exchange's constructor throws it; bench.py checks its own line with it.
std::string group_error(int world, int comm_ranks, const std::string &devices_csv);

// Test hook: B9_TEST_STALL="<where>:<rank>" parks that rank for ever at the named point ("start": before anything touches
// a GPU; "before-init" / "after-init": around ncclCommInitRank), so the launchers' deadlines can be exercised.
void test_stall(const char *where, int rank);

}  // namespace b9h
