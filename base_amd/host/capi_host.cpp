// capi_host.cpp -- the C surface of libbase9host.so declared in include/base9_host.h: the walker-parallel sampler and
// its exchange for callers that are not C++ (bench.py, tests), and the host loaders so that tests can check the parsers
// against the generators without a GPU.  No numerics.
#include "../../include/base9_host.h"
#include "b9host.hpp"
#include "b9sampler.hpp"
#include "cli_common.hpp"

#include <hip/hip_runtime_api.h>

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

namespace {
std::string g_err;
template <class F> int guard(F f) { try { f(); return 0; } catch (const std::exception &e) { g_err = e.what(); return -1; } }
}

extern "C" {

const char *b9h_last_error(void) { return g_err.c_str(); }

// Loads a pack and hands out a b9_pack view; the storage lives until b9h_free_pack.
int b9h_load_pack(const char *dir, const char *ms_model, const char *wd_model, const char *filters_csv,
                  void **handle, b9_pack *view)
{
    return guard([&] {
        std::vector<std::string> filters;
        std::string cur;
        for (const char *p = filters_csv; ; ++p) {
            if (*p == ',' || *p == '\0') { if (!cur.empty()) filters.push_back(cur); cur.clear(); if (!*p) break; }
            else cur += *p;
        }
        auto *pk = new b9h::ModelPack(b9h::load_model_pack(dir, ms_model, wd_model, filters));
        *handle = pk;
        *view = pk->view();
    });
}
void b9h_free_pack(void *handle) { delete static_cast<b9h::ModelPack *>(handle); }

int b9h_read_phot(const char *path, double min_mag, double max_mag, int index, void **handle, b9_stars *view,
                  char *filters_out, int filters_cap)
{
    return guard([&] {
        auto *ph = new b9h::Photometry(b9h::read_photometry(path, min_mag, max_mag, index));
        *handle = ph;
        *view = ph->view();
        std::string csv;
        for (auto &f : ph->filters) csv += (csv.empty() ? "" : ",") + f;
        std::strncpy(filters_out, csv.c_str(), (size_t)filters_cap - 1);
        filters_out[filters_cap - 1] = '\0';
    });
}
void b9h_free_phot(void *handle) { delete static_cast<b9h::Photometry *>(handle); }

// Resolves settings (YAML + flags given as one string per argv entry) and returns "key = value" lines.
int b9h_settings_dump(int argc, char **argv, char *out, int cap)
{
    return guard([&] {
        b9h::Settings st;
        st.parse_args(argc, argv);
        std::string d = st.dump();
        std::strncpy(out, d.c_str(), (size_t)cap - 1);
        out[cap - 1] = '\0';
    });
}

int b9h_merge_parts(const char *final_path, int world, int walkers_per_rank, long rows_per_part)
{
    return guard([&] { b9h::merge_result_parts(final_path, world, walkers_per_rank, rows_per_part); });
}

// ---- ranks, exchange ---------------------------------------------------------------------------------------------
void b9h_rank_from_env(int *rank, int *world, int *local_rank) { b9h::rank_from_env(*rank, *world, *local_rank); }

int b9h_device_synchronize(void)
{
    return guard([&] { if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("hipDeviceSynchronize failed"); });
}

int b9h_exchange_local(void **out) { return guard([&] { *out = b9h::make_local_exchange().release(); }); }
int b9h_exchange_rccl(int rank, int world, const char *dir, int device, void **out)
{
    return guard([&] { *out = b9h::make_rccl_exchange(rank, world, dir && *dir ? std::string(dir) : b9h::default_bootstrap_dir(), device).release(); });
}
int b9h_exchange_callback(b9h_gather_fn gather, void *user, int rank, int world, void **out)
{
    return guard([&] { *out = b9h::make_callback_exchange(gather, user, rank, world).release(); });
}
void b9h_exchange_free(void *e) { delete static_cast<b9h::Exchange *>(e); }
int b9h_exchange_barrier(void *e) { return guard([&] { static_cast<b9h::Exchange *>(e)->barrier(); }); }
int b9h_exchange_max(void *e, double v, double *out) { return guard([&] { *out = static_cast<b9h::Exchange *>(e)->all_reduce_max(v); }); }
int b9h_exchange_world(void *e) { return static_cast<b9h::Exchange *>(e)->world(); }
const char *b9h_exchange_name(void *e) { return static_cast<b9h::Exchange *>(e)->name(); }
int b9h_exchange_comm_ranks(void *e) { return static_cast<b9h::Exchange *>(e)->comm_ranks(); }
int b9h_exchange_devices(void *e, char *out, int cap)
{
    const std::string d = static_cast<b9h::Exchange *>(e)->devices();
    if (!out || cap < 1 || (int)d.size() + 1 > cap) return -1;
    std::memcpy(out, d.c_str(), d.size() + 1);
    return 0;
}
int b9h_forced_ranks(void) { return b9h::forced_ranks() ? 1 : 0; }
int b9h_group_check(int world, int comm_ranks, const char *devices_csv, char *msg, int cap)
{
    const std::string bad = b9h::group_error(world, comm_ranks, devices_csv ? devices_csv : "");
    if (msg && cap > 0) { std::strncpy(msg, bad.c_str(), (size_t)cap - 1); msg[cap - 1] = '\0'; }
    return bad.empty() ? 0 : 1;
}
void b9h_test_stall(const char *where, int rank) { b9h::test_stall(where, rank); }

// ---- sampler --------------------------------------------------------------------------------------------------------
namespace {
struct SamplerBox {
    std::unique_ptr<b9h::BlockRunner> runner;
    std::unique_ptr<b9h::WalkerSampler> sampler;
};
b9h::SamplerConfig make_config(int n_walkers, const int32_t *free_idx, const double *step, int d, uint64_t seed, int block)
{
    b9h::SamplerConfig c;
    c.n_walkers = n_walkers; c.free_idx.assign(free_idx, free_idx + d); c.step.assign(step, step + d); c.seed = seed; c.block = block;
    return c;
}
std::vector<int32_t> local_ids(int n_walkers, b9h::Exchange *ex)
{
    if (n_walkers < 1 || n_walkers % ex->world()) throw std::runtime_error("the number of walkers must be a multiple of the number of ranks");
    const int per = n_walkers / ex->world();
    std::vector<int32_t> ids(per);
    for (int k = 0; k < per; ++k) ids[k] = ex->rank() * per + k;
    return ids;
}
}  // namespace

int b9h_sampler_create(b9_ctx *ctx, int mode, int n_walkers, const int32_t *free_idx, const double *step, int d, uint64_t seed,
                       int block, void *exchange, void **out)
{
    return guard([&] {
        auto *ex = static_cast<b9h::Exchange *>(exchange);
        const b9h::SamplerConfig cfg = make_config(n_walkers, free_idx, step, d, seed, block);
        const std::vector<int32_t> ids = local_ids(n_walkers, ex);
        auto box = std::make_unique<SamplerBox>();
        box->runner = b9h::make_device_runner(ctx, (int)ids.size(), ids, cfg.free_idx, seed, mode);
        box->sampler = std::make_unique<b9h::WalkerSampler>(cfg, box->runner.get(), ex);
        *out = box.release();
    });
}
int b9h_sampler_create_callback(b9h_block_fn run, b9h_logpost_fn eval, void *user, int n_walkers, const int32_t *free_idx,
                                const double *step, int d, uint64_t seed, int block, void *exchange, void **out)
{
    return guard([&] {
        auto *ex = static_cast<b9h::Exchange *>(exchange);
        const b9h::SamplerConfig cfg = make_config(n_walkers, free_idx, step, d, seed, block);
        const std::vector<int32_t> ids = local_ids(n_walkers, ex);
        auto box = std::make_unique<SamplerBox>();
        box->runner = b9h::make_callback_runner(run, eval, user, (int)ids.size(), ids, cfg.free_idx, seed);
        box->sampler = std::make_unique<b9h::WalkerSampler>(cfg, box->runner.get(), ex);
        *out = box.release();
    });
}
void b9h_sampler_free(void *s) { delete static_cast<SamplerBox *>(s); }
int b9h_sampler_initialise(void *s, const double *start) { return guard([&] { static_cast<SamplerBox *>(s)->sampler->initialise(start); }); }
int b9h_sampler_run(void *s, int64_t n_steps, int adapt, double *samples, double *lps)
{
    return guard([&] {
        b9h::WalkerSampler &sm = *static_cast<SamplerBox *>(s)->sampler;
        const long first = sm.steps();
        b9h::RecordFn rec;
        if (samples || lps)
            rec = [&](const b9h::BlockRecord &r) {
                const size_t row = (size_t)(r.step0 - first) * r.n_local;
                if (samples) std::memcpy(samples + row * r.d, r.samples, sizeof(double) * (size_t)r.n_steps * r.n_local * r.d);
                if (lps) std::memcpy(lps + row, r.lps, sizeof(double) * (size_t)r.n_steps * r.n_local);
            };
        sm.run((long)n_steps, adapt != 0, rec);
    });
}
int b9h_sampler_n_local(void *s) { return static_cast<SamplerBox *>(s)->sampler->n_local(); }
int b9h_sampler_state(void *s, int64_t *steps, int64_t *accepted_local, double *scale, double *chol, double *all_params, double *all_logpost)
{
    return guard([&] {
        const b9h::WalkerSampler &sm = *static_cast<SamplerBox *>(s)->sampler;
        if (steps) *steps = sm.steps();
        if (accepted_local) *accepted_local = sm.accepted_local();
        if (scale) *scale = sm.scale();
        if (chol) std::copy(sm.chol().begin(), sm.chol().end(), chol);
        if (all_params) std::copy(sm.all_params().begin(), sm.all_params().end(), all_params);
        if (all_logpost) std::copy(sm.all_logpost().begin(), sm.all_logpost().end(), all_logpost);
    });
}
int b9h_summary_rows(const double *samples, const double *params_end, const double *logpost_end, int n_steps, int n_local, int d,
                     const double *origin, double *rows)
{
    return guard([&] { b9h::summary_rows(samples, params_end, logpost_end, n_steps, n_local, d, origin, rows); });
}

}  // extern "C"
