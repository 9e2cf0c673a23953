// b9_capi_eval.cpp -- one log-posterior evaluation through the C ABI: derive -> stars -> finalize (b9_logpost,
// b9_logpost_device), the per-star mass draws (b9_sample_mass) and the isochrone dump (b9_derive_isochrone).
#include "b9_ctx.h"
#include <atomic>

using namespace b9i;

namespace b9i {

// ping-pong work-buffer set (0 / 1)
Bufs buffer_set(const b9_ctx *ctx, int set)
{
    const size_t rows = (size_t)ctx->cap_walkers * ctx->cap_pops;
    return Bufs{ctx->d_params + (size_t)set * ctx->cap_walkers * B9_NPARAM, ctx->d_hdr + (size_t)set * rows,
                ctx->d_iso + (size_t)set * rows * ctx->iso_stride};
}

// number of partial sums one walker gets from the star kernel under the current plan / mode
// (marginalised mode: one per 64-star chunk -- the star kernel sums a chunk's values in a fixed order -- and one per four WD-stage stars)
int partial_count(const b9_ctx *ctx, const B9Groups &plan)
{
    return ctx->opt.mode == B9_MODE_MARGINALISED ? ctx->st.mg_pad / 64 + (ctx->st.n_wd + 3) / 4 : plan.n_groups * 4 + ctx->heavy_parts;
}

// doubles between two walkers' partial rows (room for either mode's row)
long long partial_stride(const b9_ctx *ctx) { return (long long)ctx->st.n_pad + ctx->st.n_pad / 64; }

// The star-likelihood launch (given-mass: hot + heavy workgroups; marginalised: one wave per star)
// on buffer set `set`, bracketed by timing events when sampled.
int launch_stars(b9_ctx *ctx, const Bufs &bf, int32_t n_walkers, double *d_perstar, const B9Groups &plan,
                        hipStream_t stream)
{
    const int n_pops = ctx->opt.n_pops;
    size_t slot = 0;
    const bool timed = ctx->timing > 0 && (ctx->launch_no++ % (unsigned)ctx->timing) == 0;
    if (timed) {
        if (ctx->ev_used == ctx->ev_start.size()) {
            hipEvent_t a, b;
            HIPCHK(ctx, hipEventCreate(&a));
            HIPCHK(ctx, hipEventCreate(&b));
            ctx->ev_start.push_back(a); ctx->ev_stop.push_back(b);
        }
        slot = ctx->ev_used++;
        if (ctx->ev_count.size() < ctx->ev_used) ctx->ev_count.resize(ctx->ev_used, 1);
        ctx->ev_count[slot] = 1;
        HIPCHK(ctx, hipEventRecord(ctx->ev_start[slot], stream));
    }
    if (ctx->opt.mode == B9_MODE_MARGINALISED) {
        const int K = ctx->opt.marg_iso_increm > 0 ? ctx->opt.marg_iso_increm : 1;
        const int Q = ctx->opt.marg_n_q > 0 ? ctx->opt.marg_n_q : 1;
        const int rc = ensure_marg_table(ctx, n_walkers, n_pops, K, Q);
        if (rc) return rc;
        HIPCHK(ctx, b9k_star_marg(ctx->pk, ctx->st, bf.hdr, bf.iso, ctx->iso_stride, ctx->mass_cap, bf.params,
                                  n_walkers, n_pops, ctx->d_partial, partial_stride(ctx), d_perstar, K, Q, nullptr, ctx->marg_prune, ctx->d_marg_tab, ctx->d_marg_wd_tab, ctx->d_marg_shares, stream));
    } else {
        HIPCHK(ctx, b9k_star_like(ctx->pk, ctx->st, bf.hdr, bf.iso, ctx->iso_stride, ctx->mass_cap, bf.params,
                                  n_walkers, n_pops, ctx->d_partial, partial_stride(ctx), d_perstar, plan, ctx->heavy_parts, stream));
    }
    if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_stop[slot], stream));
    return B9_OK;
}

}  // namespace b9i

namespace {

// One log-posterior evaluation of rows that are already in buffer set 0's parameter rows (or in
// d_params when that is a caller's device pointer): derive -> stars -> finalize.
int launch_logpost(b9_ctx *ctx, double *d_params, int32_t n_walkers, double *d_logpost,
                          double *d_perstar, hipStream_t stream, const double *host_rows = nullptr,
                          unsigned long long *done_flag = nullptr, unsigned long long done_seq = 0)
{
    const int n_pops = ctx->opt.n_pops;
    const B9Groups plan = make_plan(ctx, n_walkers, n_pops);
    int rc = ensure_capacity(ctx, n_walkers, n_pops, (size_t)partial_stride(ctx) * n_walkers, false);
    if (rc) return rc;
    Bufs bf = buffer_set(ctx, 0);
    bf.params = d_params;
    const McmcDev off{};
    const B9Prev none{nullptr, 0, 0, nullptr, nullptr};
    if (host_rows)      // <= 8 rows travel in the kernel arguments: no upload
        HIPCHK(ctx, b9k_derive_iso_rows(ctx->pk, host_rows, bf.params, n_walkers, n_pops, bf.hdr, bf.iso, ctx->iso_stride,
                                        ctx->mass_cap, stream));
    else
        HIPCHK(ctx, b9k_derive_iso(ctx->pk, bf.params, n_walkers, n_pops, bf.hdr, bf.iso, ctx->iso_stride, ctx->mass_cap,
                                   off, ctx->pr, none, stream));
    rc = launch_stars(ctx, bf, n_walkers, d_perstar, plan, stream);
    if (rc) return rc;
    HIPCHK(ctx, b9k_finalize(bf.hdr, ctx->d_partial, partial_count(ctx, plan), partial_stride(ctx), n_pops, bf.params, ctx->pr,
                             n_walkers, d_logpost, d_perstar, ctx->st.n, off, stream, done_flag, done_seq));
    return B9_OK;
}

}  // namespace

extern "C" {

int b9_logpost_device(b9_ctx *ctx, const double *d_params, int32_t n_walkers, double *d_logpost,
                      double *d_perstar, void *stream_v)
{
    if (!ctx || !d_params || !d_logpost || n_walkers < 1) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    int rc = check_ready(ctx);
    if (rc) return rc;
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : ctx->stream;
    return launch_logpost(ctx, const_cast<double *>(d_params), n_walkers, d_logpost, d_perstar, stream);
}

int b9_logpost(b9_ctx *ctx, const double *params, int32_t n_walkers, double *out_logpost, double *out_perstar)
{
    if (!ctx || !params || !out_logpost || n_walkers < 1) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    int rc = check_ready(ctx);
    if (rc) return rc;
    const B9Groups plan = make_plan(ctx, n_walkers, ctx->opt.n_pops);
    (void)plan;
    rc = ensure_capacity(ctx, n_walkers, ctx->opt.n_pops, (size_t)partial_stride(ctx) * n_walkers, out_perstar != nullptr);
    if (rc) return rc;
    // The per-step call of a host-driven sampler (the reference's logPostStep) is latency: for up
    // to 8 rows the parameters ride in the first launch's kernel arguments and the log-posteriors are written by
    // k_finalize straight into pinned host memory mapped into the device -- no copy command in the stream at all.
    if (!ctx->h_lp) {
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_lp, sizeof(double) * 16, hipHostMallocMapped));
        HIPCHK(ctx, hipHostGetDevicePointer((void **)&ctx->h_lp_dev, ctx->h_lp, 0));
        std::memset(ctx->h_lp, 0, sizeof(double) * 16);
    }
    const bool small = n_walkers <= 8;
    // ... and the host does not wait for the stream's completion signal either (a wake-up of several microseconds): the
    // last launch stores a per-call sequence number behind every log-posterior and the host polls those words
    volatile unsigned long long *const h_done = reinterpret_cast<volatile unsigned long long *>(ctx->h_lp + 8);
    const bool polled = small && !out_perstar;
    const unsigned long long seq = ++ctx->lp_seq;
    if (small) {
        rc = launch_logpost(ctx, ctx->d_params, n_walkers, ctx->h_lp_dev, out_perstar ? ctx->d_perstar : nullptr, ctx->stream, params,
                            polled ? reinterpret_cast<unsigned long long *>(ctx->h_lp_dev + 8) : nullptr, seq);
        if (rc) return rc;
    } else {
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_params, params, sizeof(double) * B9_NPARAM * n_walkers, hipMemcpyHostToDevice, ctx->stream));
        rc = b9_logpost_device(ctx, ctx->d_params, n_walkers, ctx->d_logpost, out_perstar ? ctx->d_perstar : nullptr, ctx->stream);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(out_logpost, ctx->d_logpost, sizeof(double) * n_walkers, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (out_perstar)
        HIPCHK(ctx, hipMemcpyAsync(out_perstar, ctx->d_perstar, sizeof(double) * (size_t)n_walkers * ctx->st.n,
                                   hipMemcpyDeviceToHost, ctx->stream));
    if (polled) {
        // (bounded: a launch that failed never stores its words -- after ~2 ms the stream's own wait takes over and reports)
        bool done = false;
        for (long spin = 0; spin < 2000000 && !done; ++spin) {
            done = true;
            for (int w = 0; w < n_walkers; ++w) done = done && h_done[w] == seq;
            if (!done) __builtin_ia32_pause();
        }
        if (!done) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (small) std::memcpy(out_logpost, ctx->h_lp, sizeof(double) * n_walkers);
    return B9_OK;
}

int b9_sample_mass(b9_ctx *ctx, const double *params, int32_t n_rows, uint64_t seed, int64_t row0,
                   double *out_mass, double *out_ratio, double *out_member, int32_t *out_pop)
{
    if (!ctx || !params || n_rows < 1 || !out_mass || !out_ratio || !out_member) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    int rc = check_ready(ctx);
    if (rc) return rc;
    const int n_pops = ctx->opt.n_pops, n = ctx->st.n;
    const int K = ctx->opt.marg_iso_increm > 0 ? ctx->opt.marg_iso_increm : 1;
    const int Q = ctx->opt.marg_n_q > 0 ? ctx->opt.marg_n_q : 1;
    const int chunk = std::min<int>(n_rows, 32);
    rc = ensure_capacity(ctx, chunk, n_pops, (size_t)partial_stride(ctx) * chunk, false);
    if (rc) return rc;
    rc = ensure_marg_table(ctx, chunk, n_pops, K, Q);
    if (rc) return rc;
    double *d_out = nullptr;
    int *d_pop = nullptr;
    const size_t per = (size_t)chunk * n;
    HIPCHK(ctx, hipMalloc((void **)&d_out, sizeof(double) * per * 3));
    if (out_pop && hipMalloc((void **)&d_pop, sizeof(int) * per) != hipSuccess) { (void)hipFree(d_out); return fail(ctx, B9_ERR_HIP, "hipMalloc failed"); }
    hipStream_t s = ctx->stream;
    const Bufs bf = buffer_set(ctx, 0);
    const McmcDev off{};
    rc = B9_OK;
    for (int r0 = 0; r0 < n_rows && rc == B9_OK; r0 += chunk) {
        const int m = std::min(chunk, n_rows - r0);
        hipError_t e = hipMemcpyAsync(bf.params, params + (size_t)r0 * B9_NPARAM, sizeof(double) * B9_NPARAM * m, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, sizeof(double) * per * 3, s);     // rows outside the grid write nothing
        if (e == hipSuccess && d_pop) e = hipMemsetAsync(d_pop, 0, sizeof(int) * per, s);
        if (e == hipSuccess) e = b9k_derive_iso(ctx->pk, bf.params, m, n_pops, bf.hdr, bf.iso, ctx->iso_stride, ctx->mass_cap, off, ctx->pr,
                                                B9Prev{nullptr, 0, 0, nullptr, nullptr}, s);
        B9MargSample smp{d_out, d_out + per, d_out + 2 * per, d_pop, (unsigned)(seed & 0xFFFFFFFFull), (unsigned)(seed >> 32), (long long)(row0 + r0)};
        // the kernel indexes its outputs [row][n_stars] with the launch's own row count: rows are contiguous for any m
        if (e == hipSuccess) e = b9k_star_marg(ctx->pk, ctx->st, bf.hdr, bf.iso, ctx->iso_stride, ctx->mass_cap, bf.params, m, n_pops,
                                               ctx->d_partial, partial_stride(ctx), nullptr, K, Q, &smp, ctx->marg_prune, ctx->d_marg_tab, ctx->d_marg_wd_tab, ctx->d_marg_shares, s);
        const size_t cnt = (size_t)m * n, o = (size_t)r0 * n;
        if (e == hipSuccess) e = hipMemcpyAsync(out_mass + o, d_out, sizeof(double) * cnt, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(out_ratio + o, d_out + per, sizeof(double) * cnt, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(out_member + o, d_out + 2 * per, sizeof(double) * cnt, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && d_pop) e = hipMemcpyAsync(out_pop + o, d_pop, sizeof(int) * cnt, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = fail(ctx, B9_ERR_HIP, std::string("b9_sample_mass: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_out);
    if (d_pop) (void)hipFree(d_pop);
    return rc;
}

int b9_derive_isochrone(b9_ctx *ctx, const double *param_row, int32_t pop, int32_t cap, double *out_mass,
                        double *out_mags, int32_t *out_first_eep, int32_t *out_n, double *out_agb_tip)
{
    if (!ctx || !param_row || !out_mass || !out_mags || !out_first_eep || !out_n || !out_agb_tip) return B9_ERR_INVALID;
    if (!ctx->have_pack) return fail(ctx, B9_ERR_STATE, "load the pack first");
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_capacity(ctx, 1, 1, 1, false);
    if (rc) return rc;
    double row[B9_NPARAM];
    std::memcpy(row, param_row, sizeof row);
    if (pop) row[B9_P_Y] = row[B9_P_Y2];
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_params, row, sizeof row, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, b9k_derive_iso(ctx->pk, ctx->d_params, 1, 1, ctx->d_hdr, ctx->d_iso, ctx->iso_stride, ctx->mass_cap, McmcDev{}, ctx->pr, B9Prev{nullptr, 0, 0, nullptr, nullptr}, ctx->stream));
    IsoHdr h;
    HIPCHK(ctx, hipMemcpyAsync(&h, ctx->d_hdr, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out_n = 0; *out_first_eep = 0; *out_agb_tip = 0.0;
    if (!h.valid) return B9_OK;
    if (h.n > cap) return fail(ctx, B9_ERR_CAPACITY, "isochrone longer than the caller's buffers");
    const int nf = ctx->pk.nf, nfp = ctx->pk.nfp;
    std::vector<double> buf((size_t)h.n * nfp);
    HIPCHK(ctx, hipMemcpy(out_mass, ctx->d_iso, sizeof(double) * h.n, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(buf.data(), ctx->d_iso + ctx->mass_cap, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
    for (int e = 0; e < h.n; ++e) std::memcpy(&out_mags[(size_t)e * nf], &buf[(size_t)e * nfp], sizeof(double) * nf);
    *out_n = h.n; *out_first_eep = h.first_eep; *out_agb_tip = h.agb_tip;
    return B9_OK;
}

}  // extern "C"
