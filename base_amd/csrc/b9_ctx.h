// b9_ctx.h -- what the translation units of the C ABI share (internal; not installed): the context, and the helpers of
// one file that another one calls.  The ABI itself is include/base9_hip.h; its implementation is
//   b9_capi_ctx.cpp     context life cycle, options / tuning, work buffers, introspection, timing
//   b9_capi_stage.cpp   validation and staging of the model pack and the star catalogue into HBM (b9_load_pack, b9_load_stars)
//   b9_capi_plan.cpp    launch plans: canonical tile groups, the fused step's and the tree step's plans
//   b9_capi_margplan.cpp  the marginalised mode's catalogue plan: measured dispatch order, pieces of small catalogues
//   b9_capi_eval.cpp    b9_logpost / b9_logpost_device / b9_sample_mass / b9_derive_isochrone
//   b9_capi_blocks.cpp  the sampler's device-resident blocks (fused, tree-speculative, two-launch), b9_mcmc_run_block / b9_mcmc_wait
#pragma once
#include "../../include/base9_hip.h"
#include "b9_device.h"
#include "b9_launch.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

namespace b9i {

struct HostStars {
    int n = 0, nf = 0;
    std::vector<double> obs, sigma, mass1, q, prior, fmin, fmax;
    std::vector<int> stage, wd_type;
    double min_mass1 = 0.0;
};

}  // namespace b9i

struct b9_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    bool have_pack = false;
    DevPack pk{};
    std::vector<void *> pack_allocs;

    bool have_stars = false, stars_dirty = false;
    b9i::HostStars hs;
    DevStars st{};
    std::vector<void *> star_allocs;

    DevPriors pr{};
    b9_options opt{B9_MODE_GIVEN_MASS, 1, 8, 8};

    // per-call work buffers (grown on demand, never shrunk)
    int cap_walkers = 0, cap_pops = 0;
    IsoHdr *d_hdr = nullptr;
    double *d_iso = nullptr;
    long long iso_stride = 0;
    int mass_cap = 0;
    double *d_partial = nullptr;
    size_t partial_cap = 0;
    double *d_params = nullptr, *d_logpost = nullptr, *d_perstar = nullptr;
    size_t perstar_cap = 0;
    double *d_marg_tab = nullptr;    // marginalised mode: the companions' flux table of the current call (k_marg_table)
    size_t marg_tab_cap = 0;
    double *d_marg_wd_tab = nullptr; // ... and the WD-stage stars' node table (k_marg_wd_table)
    size_t marg_wd_tab_cap = 0;
    double *d_marg_shares = nullptr; // ... and the per-star shares of a split k_star_marg launch (small catalogues)
    size_t marg_shares_cap = 0;
    // the marginalised mode's catalogue plan (b9_capi_margplan.cpp): measured dispatch order and pieces; remade after the stars,
    // the pack, the priors or the options change
    bool marg_plan_ok = false;
    int marg_piece_units = 0;                 // b9_tuning.marg_piece_units: 0 = default
    std::vector<void *> marg_plan_allocs;
    const int *marg_order_spread = nullptr;   // the load-time order (photometric spread), owned by star_allocs
    std::vector<double> marg_cost;            // measured cost per star chunk (units; empty: not measured)
    std::vector<double> h_log_age, h_feh, h_y;   // host copies of the pack's grid axes (the plan's reference row is clamped into them)
    struct McmcSlot {                // fused step: one enqueued block (device block, pinned mirror, completion event)
        void *d = nullptr, *h = nullptr, *h_dev = nullptr;   // h_dev: the pinned mirror as the device sees it (mapped)
        size_t cap = 0, hcap = 0;
        hipEvent_t done = nullptr;
        bool in_flight = false;
        const void *owner = nullptr; // the b9_mcmc_block it was enqueued for
        int W = 0, final_parity = 0;
        size_t o_nacc = 0, o_st0 = 0, o_st1 = 0, o_samp = 0, o_lps = 0, o_rows = 0, n_samp = 0, n_lps = 0, n_rows = 0;
        bool host_samples = false; // the caller asked for the chain record (else it only exists on the device, for the rows)
        hipEvent_t rows_ready = nullptr;   // recorded right after the block's last kernel: the summary rows are in HBM
        int kind = 0;                      // 0: fused one-launch steps; 1: two-launch steps (marginalised mode)
        size_t o_cur = 0, o_lp = 0;        // two-launch blocks: where the final state half sits in the block
    } slot[2];
    int next_slot = 0, last_slot = -1;
    double *h_lp = nullptr, *h_lp_dev = nullptr;   // b9_logpost: 8 log-posteriors + 8 completion words in mapped pinned host memory (host / device view)
    unsigned long long lp_seq = 0;                 // ... and the number of the call the completion words announce

    // launch plan
    int n_cu = 256;            // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    int plan_debug_key = -1;
    int step_blocks_per_cu = 0, step_occ_key = -1;   // k_mcmc_step workgroups per CU for (nfp, n_pops, mass_cap), and the key it was queried for
    int heavy_parts = 4;       // workgroups per walker for the stars above the AGB tip (sized in check_ready)
    int n_wd_stage = 0;        // stars the catalogue marks as white dwarfs
    b9_tuning tuning{};        // the tuning in force (b9_get_tuning): the environment's at creation, then the last b9_set_tuning
    int tiles_per_block = 0;   // 0 = auto
    int derive_parts = 0;      // fused sampler step: workgroups per candidate isochrone (0 = one value per thread)
    int derive_order = 1;      // fused sampler step: 1 writers + derivation lead the grid and the heavy-star workgroups follow them (default),
                               // 0 heavy-star workgroups first, < 0 derivation workgroups trail the hot ones (B9_DERIVE_ORDER)
    bool two_launch_steps = false;   // b9_tuning.two_launch_steps: the derive + star launch pair per step also in given-mass mode
    int plan_debug = 0;              // b9_tuning.plan_debug: print the fused step's launch plan to stderr when it changes (2: the marginalised catalogue's pieces too)
    bool marg_prune = true;          // marginalised kernel: field floor + box pruning (b9_tuning.marg_no_pruning turns both off)
    int heavy_parts_fixed = 0;       // b9_tuning.heavy_parts: 0 = sized from the catalogue (check_ready)
    int tree_depth = 0;              // b9_tuning.tree_depth: 0 = automatic
    int tree_blocks_per_cu = 0, tree_occ_key = -1;   // k_mcmc_tree workgroups per CU, and the key it was queried for
    // candidate buffers of the tree-speculative step (grown on demand): [2 parities][W][outcomes][nodes]([pops])
    IsoHdr *d_tree_hdr = nullptr;
    double *d_tree_iso = nullptr, *d_tree_par = nullptr, *d_tree_partial = nullptr;
    size_t tree_cand_cap = 0, tree_partial_cap = 0;
    long long tree_iso_stride = 0;

    // timing of the dominant kernel
    int timing = 0;            // 0 off, n > 0: bracket every n-th launch of the dominant kernel with events
    unsigned long long launch_no = 0;
    std::vector<hipEvent_t> ev_start, ev_stop;
    size_t ev_used = 0;
    std::vector<int> ev_count;      // launches covered by each bracket
    int timing_group = 8;            // fused step: a bracket spans this many consecutive launches (B9_TIMING_GROUP)
    double ms_accum = 0.0;
    int launches = 0;
    unsigned long long *d_clock = nullptr;   // b9_clock_stamp: [2 stamps][B9_CLOCK_SLOTS]{s_memtime, s_memrealtime}
};

namespace b9i {

inline int fail(b9_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, B9_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

// an enqueued sampler block owns the context's work buffers (candidate isochrones, partial sums) until it is collected
inline bool block_outstanding(const b9_ctx *ctx)
{
    for (const auto &sl : ctx->slot) if (sl.in_flight) return true;
    return false;
}
constexpr const char *kBlockOutstanding = "a sampler block is outstanding: collect it with b9_mcmc_wait first (it owns the context's work buffers)";

inline void free_all(std::vector<void *> &v)
{
    for (void *p : v) (void)hipFree(p);
    v.clear();
}

// ---- b9_capi_stage.cpp
int build_stars(b9_ctx *ctx);

// ---- b9_capi_ctx.cpp
int ensure_capacity(b9_ctx *ctx, int n_walkers, int n_pops, size_t n_partial, bool want_perstar);
int ensure_marg_table(b9_ctx *ctx, int n_walkers, int n_pops, int K, int Q);
int check_ready(b9_ctx *ctx);
int timing_begin(b9_ctx *ctx, hipStream_t stream, long *slot);
int timing_end(b9_ctx *ctx, hipStream_t stream, long slot);

// ---- b9_capi_margplan.cpp
int ensure_marg_plan(b9_ctx *ctx);

// ---- b9_capi_plan.cpp
struct Groups { int group_tiles, n_groups; };
struct StepPlan { B9Groups plan; int derive_parts; };
struct TreePlan { int depth, group_tiles, n_groups, derive_parts; };
B9Groups make_plan(b9_ctx *ctx, int n_walkers, int n_pops);
StepPlan make_step_plan(b9_ctx *ctx, int n_walkers, int n_pops);
TreePlan make_tree_plan(b9_ctx *ctx, int n_walkers, int n_pops);
int ensure_tree_buffers(b9_ctx *ctx, int n_walkers, int n_pops, const TreePlan &tp);
void apply_tuning(b9_ctx *ctx, const b9_tuning &t);
bool tuning_from_env(b9_tuning *t);

// ---- b9_capi_eval.cpp
struct Bufs { double *params; IsoHdr *hdr; double *iso; };
Bufs buffer_set(const b9_ctx *ctx, int set);
int partial_count(const b9_ctx *ctx, const B9Groups &plan);
long long partial_stride(const b9_ctx *ctx);
int launch_stars(b9_ctx *ctx, const Bufs &bf, int32_t n_walkers, double *d_perstar, const B9Groups &plan, hipStream_t stream);

}  // namespace b9i
