// b9_capi_ctx.cpp -- context life cycle, options / tuning, the per-call work buffers, introspection and the HIP-event
// timing of the dominant kernel (include/base9_hip.h).
#include "b9_ctx.h"

using namespace b9i;

namespace {

std::string g_create_error;

}  // namespace

namespace b9i {

int ensure_capacity(b9_ctx *ctx, int n_walkers, int n_pops, size_t n_partial, bool want_perstar)
{
    // (the isochrone rows depend on BOTH the longest isochrone and the padded filter count of the loaded pack: a pack
    //  reloaded with the same EEP count but more filters needs wider rows)
    const int want_cap = (ctx->pk.max_eep + 1) & ~1;
    if (n_walkers > ctx->cap_walkers || n_pops > ctx->cap_pops || ctx->mass_cap != want_cap ||
        ctx->iso_stride != (long long)want_cap * (ctx->pk.nfp + 1)) {
        if (ctx->d_hdr) (void)hipFree(ctx->d_hdr);
        if (ctx->d_iso) (void)hipFree(ctx->d_iso);
        if (ctx->d_params) (void)hipFree(ctx->d_params);
        if (ctx->d_logpost) (void)hipFree(ctx->d_logpost);
        ctx->d_hdr = nullptr; ctx->d_iso = nullptr; ctx->d_params = nullptr; ctx->d_logpost = nullptr;
        int cw = std::max(n_walkers, ctx->cap_walkers), cp = std::max(n_pops, ctx->cap_pops);
        ctx->mass_cap = (ctx->pk.max_eep + 1) & ~1;
        ctx->iso_stride = (long long)ctx->mass_cap * (ctx->pk.nfp + 1);
        // four sets: the two-launch sampler (marginalised mode) ping-pongs between sets 0 and 1; the fused
        // sampler step (given-mass mode) keeps two candidates for each of two step parities (StepDev)
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_hdr, sizeof(IsoHdr) * cw * cp * 4));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_iso, sizeof(double) * (size_t)ctx->iso_stride * cw * cp * 4));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_params, sizeof(double) * B9_NPARAM * cw * 4));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_logpost, sizeof(double) * cw));
        ctx->cap_walkers = cw; ctx->cap_pops = cp;
    }
    if (n_partial > ctx->partial_cap) {
        if (ctx->d_partial) (void)hipFree(ctx->d_partial);
        ctx->d_partial = nullptr;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_partial, sizeof(double) * n_partial));
        ctx->partial_cap = n_partial;
    }
    if (want_perstar) {
        size_t need = (size_t)n_walkers * ctx->st.n;
        if (need > ctx->perstar_cap) {
            if (ctx->d_perstar) (void)hipFree(ctx->d_perstar);
            ctx->d_perstar = nullptr;
            HIPCHK(ctx, hipMalloc((void **)&ctx->d_perstar, sizeof(double) * std::max<size_t>(need, 1)));
            ctx->perstar_cap = need;
        }
    }
    return B9_OK;
}


// the marginalised mode's per-call node table for n_walkers rows (grown on demand)
int ensure_marg_table(b9_ctx *ctx, int n_walkers, int n_pops, int K, int Q)
{
    const size_t need = (size_t)n_walkers * n_pops * (size_t)b9k_marg_table_doubles(ctx->pk.nfp, ctx->mass_cap, K, Q);
    if (need > ((size_t)8 << 30) / sizeof(double)) return fail(ctx, B9_ERR_CAPACITY, "marginalisation grid too fine: the node table would exceed 8 GiB");
    if (need > ctx->marg_tab_cap) {
        if (ctx->d_marg_tab) (void)hipFree(ctx->d_marg_tab);
        ctx->d_marg_tab = nullptr; ctx->marg_tab_cap = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_marg_tab, need * sizeof(double)));
        ctx->marg_tab_cap = need;
    }
    const size_t need_wd = (size_t)n_walkers * n_pops * (size_t)b9k_marg_wd_table_doubles(ctx->pk.nfp, K);
    if (need_wd > ctx->marg_wd_tab_cap) {
        if (ctx->d_marg_wd_tab) (void)hipFree(ctx->d_marg_wd_tab);
        ctx->d_marg_wd_tab = nullptr; ctx->marg_wd_tab_cap = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_marg_wd_tab, need_wd * sizeof(double)));
        ctx->marg_wd_tab_cap = need_wd;
    }
    const size_t need_sh = (size_t)n_walkers * (size_t)b9k_marg_shares_doubles(ctx->st.mg_n_pieces, n_pops);
    if (need_sh > ctx->marg_shares_cap) {
        if (ctx->d_marg_shares) (void)hipFree(ctx->d_marg_shares);
        ctx->d_marg_shares = nullptr; ctx->marg_shares_cap = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_marg_shares, need_sh * sizeof(double)));
        ctx->marg_shares_cap = need_sh;
    }
    return B9_OK;
}

int check_ready(b9_ctx *ctx)
{
    if (!ctx->have_pack || !ctx->have_stars) return fail(ctx, B9_ERR_STATE, "load the pack and the stars first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->stars_dirty) { int rc = build_stars(ctx); if (rc) return rc; }
    {   // Workgroups per walker for the stars above the AGB tip (WD branch / NS-BH).  Their number depends on the walker's
        // age; the catalogue's WD-stage stars plus 2 % of the rest is the estimate (1 % with two populations: what lies
        // between the two populations' tips is a thinner slice, and the launch has fewer slots to spare -- 30k stars x 2
        // populations x 8 walkers: 5 parts 20.6 us per step, 10 parts 22.0).  A star takes a lane pair of its population's
        // two (one population: four) waves and the role is a latency chain, so there is one workgroup per 64 n_pops stars.
        const int est = (ctx->n_wd_stage + ctx->hs.n / (50 * ctx->opt.n_pops)) * 2 * ctx->opt.n_pops;
        ctx->heavy_parts = std::max(4, std::min(16, (est + 255) / 256));
        if (ctx->heavy_parts_fixed > 0) ctx->heavy_parts = std::max(1, std::min(64, ctx->heavy_parts_fixed));
    }
    if (ctx->opt.mode == B9_MODE_GIVEN_MASS && ctx->hs.min_mass1 <= 0.0)
        return fail(ctx, B9_ERR_INVALID, "given-mass mode needs mass1 > 0 for every star (the marginalised mode takes mass1 as a hint only)");
    return ensure_marg_plan(ctx);
}

// event bracket of the dominant kernel's launch, every ctx->timing-th launch
int timing_begin(b9_ctx *ctx, hipStream_t stream, long *slot)
{
    *slot = -1;
    if (!(ctx->timing > 0 && (ctx->launch_no++ % (unsigned)ctx->timing) == 0)) return B9_OK;
    if (ctx->ev_used == ctx->ev_start.size()) {
        hipEvent_t a, b;
        HIPCHK(ctx, hipEventCreate(&a));
        HIPCHK(ctx, hipEventCreate(&b));
        ctx->ev_start.push_back(a); ctx->ev_stop.push_back(b);
    }
    *slot = (long)ctx->ev_used++;
    if (ctx->ev_count.size() < ctx->ev_used) ctx->ev_count.resize(ctx->ev_used, 1);
    ctx->ev_count[*slot] = 1;
    HIPCHK(ctx, hipEventRecord(ctx->ev_start[*slot], stream));
    return B9_OK;
}

int timing_end(b9_ctx *ctx, hipStream_t stream, long slot)
{
    if (slot >= 0) HIPCHK(ctx, hipEventRecord(ctx->ev_stop[slot], stream));
    return B9_OK;
}

}  // namespace b9i

extern "C" {

int b9_abi_version(void) { return B9_ABI_VERSION; }

int b9_ctx_create(int device_id, b9_ctx **out)
{
    if (!out) return B9_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                         " (the hot path has no CPU fallback)";
        return B9_ERR_NO_DEVICE;
    }
    if (device_id < 0) {
        if (hipGetDevice(&device_id) != hipSuccess) device_id = 0;
    }
    if (device_id >= count) { g_create_error = "device id out of range"; return B9_ERR_INVALID; }
    if (hipSetDevice(device_id) != hipSuccess) { g_create_error = "hipSetDevice failed"; return B9_ERR_NO_DEVICE; }
    b9_ctx *ctx = new b9_ctx();
    ctx->device = device_id;
    if (hipDeviceGetAttribute(&ctx->n_cu, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || ctx->n_cu < 1) ctx->n_cu = 256;
    // The context's stream has the LOWEST priority: a sampler block is a long train of short kernels, and a
    // collective a multi-GPU driver issues on its own stream (RCCL all-gather of the previous block's rows) must
    // get in at the next kernel boundary instead of waiting behind the whole train (measured with a 1-rank RCCL
    // group: the gather took 1.4 ms = the rest of the block; B9_STREAM_PRIORITY=default restores the default).
    int least = 0, greatest = 0;
    const char *prio = getenv("B9_STREAM_PRIORITY");       // (read here: the stream is made before any b9_set_tuning could run)
    const bool low = !(prio && std::string(prio) == "default") && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
    const hipError_t se = low ? hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, least)
                              : hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete ctx; g_create_error = "hipStreamCreate failed"; return B9_ERR_HIP;
    }
    for (int k = 0; k < 12; ++k) { ctx->pr.mean[k] = 0.0; ctx->pr.var[k] = 0.0; }
    ctx->pr.log_age_min = -INFINITY; ctx->pr.log_age_max = INFINITY;
    {   // environment overrides of the launch-plan tuning, read ONCE, here (include/base9_hip.h: b9_tuning documents them)
        b9_tuning t{};
        if (tuning_from_env(&t)) apply_tuning(ctx, t);
    }
    *out = ctx;
    return B9_OK;
}

void b9_ctx_destroy(b9_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_all(ctx->pack_allocs);
    free_all(ctx->star_allocs);
    free_all(ctx->marg_plan_allocs);
    void *bufs[] = {ctx->d_hdr, ctx->d_iso, ctx->d_partial, ctx->d_params, ctx->d_logpost, ctx->d_perstar, ctx->d_marg_tab, ctx->d_marg_wd_tab, ctx->d_marg_shares,
                    ctx->d_tree_hdr, ctx->d_tree_iso, ctx->d_tree_par, ctx->d_tree_partial};
    for (void *p : bufs) if (p) (void)hipFree(p);
    for (auto &sl : ctx->slot) {
        if (sl.d) (void)hipFree(sl.d);
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.rows_ready) (void)hipEventDestroy(sl.rows_ready);
    }
    if (ctx->h_lp) (void)hipHostFree(ctx->h_lp);
    if (ctx->d_clock) (void)hipFree(ctx->d_clock);
    for (auto e : ctx->ev_start) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_stop) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *b9_last_error(const b9_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int b9_set_priors(b9_ctx *ctx, const b9_priors *p)
{
    if (!ctx || !p) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    for (int k = 0; k < 12; ++k) { ctx->pr.mean[k] = p->mean[k]; ctx->pr.var[k] = p->var[k]; }
    ctx->pr.log_age_min = p->log_age_min; ctx->pr.log_age_max = p->log_age_max;
    ctx->marg_plan_ok = false;         // (the marginalised catalogue plan is measured at the prior means)
    return B9_OK;
}

int b9_set_tuning(b9_ctx *ctx, const b9_tuning *t)
{
    if (!ctx) return B9_ERR_INVALID;
    for (auto &sl : ctx->slot)
        if (sl.in_flight) return fail(ctx, B9_ERR_STATE, "b9_set_tuning: a block is outstanding");
    b9_tuning z{};
    apply_tuning(ctx, t ? *t : z);
    return B9_OK;
}

int b9_get_tuning(const b9_ctx *ctx, b9_tuning *out)
{
    if (!ctx || !out) return B9_ERR_INVALID;
    *out = ctx->tuning;
    return B9_OK;
}

int b9_set_options(b9_ctx *ctx, const b9_options *o)
{
    if (!ctx || !o) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    if (o->mode != B9_MODE_GIVEN_MASS && o->mode != B9_MODE_MARGINALISED) return fail(ctx, B9_ERR_INVALID, "unknown mode");
    if (o->n_pops != 1 && o->n_pops != 2) return fail(ctx, B9_ERR_INVALID, "n_pops must be 1 or 2");
    ctx->opt = *o;
    ctx->marg_plan_ok = false;
    return B9_OK;
}

int b9_max_eep(const b9_ctx *ctx) { return (ctx && ctx->have_pack) ? ctx->pk.max_eep : 0; }
int b9_device_id(const b9_ctx *ctx) { return ctx ? ctx->device : -1; }

int b9_bytes_per_star_eval(const b9_ctx *ctx)
{
    if (!ctx || !ctx->have_pack) return 0;
    // obs + 1/sigma^2 per (real) filter, mass1, q, c0, la (8 B each), flags (4 B)
    return 16 * ctx->pk.nf + 4 * 8 + 4;
}

int b9_step_tiles_per_block(b9_ctx *ctx, int32_t n_walkers)
{
    if (!ctx || n_walkers < 1) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    int rc = check_ready(ctx);
    if (rc) return rc;
    rc = ensure_capacity(ctx, n_walkers, ctx->opt.n_pops, (size_t)partial_stride(ctx) * n_walkers, false);   // (the plan keys on mass_cap)
    if (rc) return rc;
    if (ctx->opt.mode == B9_MODE_GIVEN_MASS && !ctx->two_launch_steps) {
        const TreePlan tp = make_tree_plan(ctx, n_walkers, ctx->opt.n_pops);
        if (tp.depth >= 2) return tp.group_tiles;
    }
    const B9Groups p = make_step_plan(ctx, n_walkers, ctx->opt.n_pops).plan;
    return p.group_tiles * p.groups_per_block;
}

int b9_step_depth(b9_ctx *ctx, int32_t n_walkers)
{
    if (!ctx || n_walkers < 1) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    int rc = check_ready(ctx);
    if (rc) return rc;
    rc = ensure_capacity(ctx, n_walkers, ctx->opt.n_pops, (size_t)partial_stride(ctx) * n_walkers, false);
    if (rc) return rc;
    if (ctx->opt.mode != B9_MODE_GIVEN_MASS || ctx->two_launch_steps) return 1;
    return make_tree_plan(ctx, n_walkers, ctx->opt.n_pops).depth;
}

int b9_enable_timing(b9_ctx *ctx, int on)
{
    if (!ctx) return B9_ERR_INVALID;
    ctx->timing = on > 0 ? on : 0;
    ctx->launch_no = 0;
    if (on > 0) {   // create the event pool now: hipEventCreate inside a timed region costs ~40 us each
        HIPCHK(ctx, hipSetDevice(ctx->device));
        while (ctx->ev_start.size() < 512) {
            hipEvent_t a, b;
            HIPCHK(ctx, hipEventCreate(&a));
            HIPCHK(ctx, hipEventCreate(&b));
            ctx->ev_start.push_back(a); ctx->ev_stop.push_back(b);
        }
    }
    return B9_OK;
}

int b9_calibrate_timing(b9_ctx *ctx, double *bracket_overhead_ms)
{
    if (!ctx || !bracket_overhead_ms) return B9_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int reps = 64;
    std::vector<hipEvent_t> ea(reps), eb(reps);
    for (int i = 0; i < reps; ++i) { HIPCHK(ctx, hipEventCreate(&ea[i])); HIPCHK(ctx, hipEventCreate(&eb[i])); }
    // Queue everything behind a ~1.5 ms spin so the brackets execute back to back, as the real
    // launches do; each has the real bracket's shape: predecessor kernel, start event, kernel, stop event.
    HIPCHK(ctx, b9k_spin(1500.0, ctx->stream));
    for (int i = 0; i < reps; ++i) {
        HIPCHK(ctx, b9k_noop(ctx->stream));
        HIPCHK(ctx, hipEventRecord(ea[i], ctx->stream));
        HIPCHK(ctx, b9k_noop(ctx->stream));
        HIPCHK(ctx, hipEventRecord(eb[i], ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (int i = 0; i < reps; ++i) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ea[i], eb[i]));
        tot += ms;
        (void)hipEventDestroy(ea[i]); (void)hipEventDestroy(eb[i]);
    }
    *bracket_overhead_ms = tot / reps;
    return B9_OK;
}

int b9_kernel_time_ms(b9_ctx *ctx, int reset, double *total_ms, int32_t *n_launches)
{
    if (!ctx || !total_ms || !n_launches) return B9_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        HIPCHK(ctx, hipEventSynchronize(ctx->ev_stop[i]));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
        ctx->ms_accum += ms;
        ctx->launches += i < ctx->ev_count.size() ? ctx->ev_count[i] : 1;
    }
    ctx->ev_used = 0;
    *total_ms = ctx->ms_accum;
    *n_launches = ctx->launches;
    if (reset) { ctx->ms_accum = 0.0; ctx->launches = 0; }
    return B9_OK;
}

int b9_clock_stamp(b9_ctx *ctx, int32_t which)
{
    if (!ctx || (which != 0 && which != 1)) return B9_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t per = (size_t)B9_CLOCK_SLOTS * 2;
    if (!ctx->d_clock) HIPCHK(ctx, hipMalloc((void **)&ctx->d_clock, sizeof(unsigned long long) * per * 2));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_clock + per * which, 0, sizeof(unsigned long long) * per, ctx->stream));
    if (which == 0) HIPCHK(ctx, hipMemsetAsync(ctx->d_clock + per, 0, sizeof(unsigned long long) * per, ctx->stream));
    HIPCHK(ctx, b9k_clock_stamp(ctx->d_clock + per * which, ctx->stream));
    return B9_OK;
}

int b9_clock_mhz(b9_ctx *ctx, double *mhz, double *mhz_min, double *mhz_max, double *ref_seconds)
{
    if (!ctx || !mhz) return B9_ERR_INVALID;
    if (!ctx->d_clock) return fail(ctx, B9_ERR_STATE, "b9_clock_mhz: no stamps (call b9_clock_stamp(ctx, 0) and (ctx, 1) first)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t per = (size_t)B9_CLOCK_SLOTS * 2;
    std::vector<unsigned long long> h(per * 2);
    HIPCHK(ctx, hipMemcpy(h.data(), ctx->d_clock, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    std::vector<double> v;
    double ref = 0.0;
    for (int x = 0; x < B9_CLOCK_SLOTS; ++x) {
        const unsigned long long t0 = h[2 * x], r0 = h[2 * x + 1], t1 = h[per + 2 * x], r1 = h[per + 2 * x + 1];
        if (!t0 || !t1 || r1 <= r0 || t1 <= t0) continue;           // no workgroup of one of the two launches landed on this CU
        v.push_back((double)(t1 - t0) / (double)(r1 - r0) * 100.0);  // s_memrealtime ticks at 100 MHz
        ref = std::max(ref, (double)(r1 - r0) * 1e-8);
    }
    if (v.empty()) return fail(ctx, B9_ERR_STATE, "b9_clock_mhz: the two stamps share no compute unit");
    std::sort(v.begin(), v.end());
    auto median = [](const std::vector<double> &x) { return x.size() % 2 ? x[x.size() / 2] : 0.5 * (x[x.size() / 2 - 1] + x[x.size() / 2]); };
    {   // a CU whose pair was torn by a pre-emption between the two counter reads shows as an outlier: dropped (3 % of the median)
        const double m0 = median(v);
        std::vector<double> kept;
        for (double x : v) if (std::fabs(x - m0) <= 0.03 * m0) kept.push_back(x);
        if (!kept.empty()) v.swap(kept);
    }
    *mhz = median(v);
    if (mhz_min) *mhz_min = v.front();
    if (mhz_max) *mhz_max = v.back();
    if (ref_seconds) *ref_seconds = ref;
    return B9_OK;
}

}  // extern "C"
