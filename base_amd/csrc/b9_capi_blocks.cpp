// b9_capi_blocks.cpp -- the sampler's device-resident Metropolis blocks (the caller of the hot path):
// fused one-launch steps (k_mcmc_step), tree-speculative launches (k_mcmc_tree), two-launch steps (marginalised mode);
// b9_mcmc_run_block enqueues one, b9_mcmc_wait collects it.
#include "b9_ctx.h"

using namespace b9i;

namespace {

/* Device-resident Metropolis block, given-mass mode: ONE launch per step (StepDev in b9_device.h).
 * Launch sequence for S steps:  D0  K(0) K(1) ... K(S-1)  F
 *   D0   = k_derive_iso: draws step 0's proposal from the starting state and derives its isochrones
 *   K(t) = k_mcmc_step: decision of step t-1, star likelihood of step t's proposal, and -- on a few
 *          extra workgroups -- both candidate isochrone sets of step t+1
 *   F    = k_mcmc_finish: decision of step S-1. */
// Collect an enqueued block: wait for its download, unpack the pinned mirror into the caller's arrays.
int collect_block(b9_ctx *ctx, b9_ctx::McmcSlot &sl, b9_mcmc_block *blk)
{
    HIPCHK(ctx, hipEventSynchronize(sl.done));
    sl.in_flight = false;
    const double *stage = static_cast<const double *>(sl.h);
    if (sl.kind == 1) {             // two-launch block: [cur][lp] of the final half, n_acc as a 64-bit count
        std::memcpy(blk->params, stage + sl.o_cur, sizeof(double) * (size_t)sl.W * B9_NPARAM);
        std::memcpy(blk->logpost, stage + sl.o_lp, sizeof(double) * (size_t)sl.W);
        if (sl.n_samp && sl.host_samples && blk->samples) std::memcpy(blk->samples, stage + sl.o_samp, sl.n_samp * 8);
        if (sl.n_rows && blk->rows) std::memcpy(blk->rows, stage + sl.o_rows, sl.n_rows * 8);
        if (sl.n_lps && blk->lps) std::memcpy(blk->lps, stage + sl.o_lps, sl.n_lps * 8);
        unsigned long long n_acc = 0;
        std::memcpy(&n_acc, stage + sl.o_nacc, sizeof n_acc);
        blk->n_accept = (int64_t)n_acc;
        return B9_OK;
    }
    const double *fin = stage + (sl.final_parity ? sl.o_st1 : sl.o_st0);
    if (sl.kind == 2) {             // tree-speculative block: tree state rows
        double n_acc = 0.0;
        for (int w = 0; w < sl.W; ++w) {
            const double *row = fin + (size_t)w * B9_TREE_STATE_STRIDE;
            std::memcpy(blk->params + (size_t)w * B9_NPARAM, row + B9_TS_CUR, sizeof(double) * B9_NPARAM);
            blk->logpost[w] = row[B9_TS_LP];
            n_acc += row[B9_TS_NACC];
        }
        if (sl.n_samp && sl.host_samples && blk->samples) std::memcpy(blk->samples, stage + sl.o_samp, sl.n_samp * 8);
        if (sl.n_rows && blk->rows) std::memcpy(blk->rows, stage + sl.o_rows, sl.n_rows * 8);
        if (sl.n_lps && blk->lps) std::memcpy(blk->lps, stage + sl.o_lps, sl.n_lps * 8);
        blk->n_accept = (int64_t)n_acc;
        return B9_OK;
    }
    for (int w = 0; w < sl.W; ++w) {
        std::memcpy(blk->params + (size_t)w * B9_NPARAM, fin + (size_t)w * B9_STATE_STRIDE + B9_ST_CUR, sizeof(double) * B9_NPARAM);
        blk->logpost[w] = fin[(size_t)w * B9_STATE_STRIDE + B9_ST_LP];
    }
    if (sl.n_samp && sl.host_samples && blk->samples) std::memcpy(blk->samples, stage + sl.o_samp, sl.n_samp * 8);
    if (sl.n_rows && blk->rows) std::memcpy(blk->rows, stage + sl.o_rows, sl.n_rows * 8);
    if (sl.n_lps && blk->lps) std::memcpy(blk->lps, stage + sl.o_lps, sl.n_lps * 8);
    double n_acc = 0.0;                              // per-walker counts carried in the state rows
    for (int w = 0; w < sl.W; ++w) n_acc += fin[(size_t)w * B9_STATE_STRIDE + B9_ST_NACC];
    blk->n_accept = (int64_t)n_acc;
    return B9_OK;
}

/* The marginalised mode runs the same block with k_marg_step in K(t)'s place (b9_marg_step.hip.h: the candidates are node
 * tables, built inside the launch; no isochrone is materialised) and, for the block's first proposal, k_marg_table behind D0. */
struct MargBlock {             // what the marginalised flavour adds to a fused block
    int K = 1, Q = 1;
    double *partial = nullptr;                 // [W][stride]: two parities of (star chunks + WD-stage stars) partials
    long long stride = 0, tab_doubles = 0, wd_stride = 0;
    int n_partial = 0;
};

// can the marginalised mode run fused steps on this context?  (the table builders' mass column must fit beside the star role's
// workgroups in LDS: isochrones of more than ~600 points at 8 filters keep the two-launch step)
bool marg_fused_ok(const b9_ctx *ctx) { return b9k_marg_step_lds(ctx->pk.nfp, ctx->mass_cap) <= B9_MSTEP_LDS_MAX(ctx->pk.nfp); }

int run_block_fused(b9_ctx *ctx, b9_mcmc_block *blk, bool marg)
{
    const int W = blk->n_walkers, d = blk->n_free, S = blk->n_steps, n_pops = ctx->opt.n_pops;
    const bool cont = (blk->flags & B9_BLOCK_CONTINUE) != 0, async = (blk->flags & B9_BLOCK_ASYNC) != 0;
    StepPlan sp{};
    MargBlock mb;
    if (marg) {
        mb.K = ctx->opt.marg_iso_increm > 0 ? ctx->opt.marg_iso_increm : 1;
        mb.Q = ctx->opt.marg_n_q > 0 ? ctx->opt.marg_n_q : 1;
        // four candidate sets (two parities x two candidates) of node tables, WD tables and split shares
        int rc = ensure_marg_table(ctx, 4 * W, n_pops, mb.K, mb.Q);
        if (rc) return rc;
        mb.n_partial = ctx->st.mg_pad / 64 + (ctx->st.n_wd + 3) / 4;
        mb.stride = 2 * (((long long)mb.n_partial + 7) & ~7ll);
        rc = ensure_capacity(ctx, W, n_pops, (size_t)std::max<long long>(mb.stride, partial_stride(ctx)) * W, false);
        if (rc) return rc;
        mb.partial = ctx->d_partial;
        mb.tab_doubles = b9k_marg_table_doubles(ctx->pk.nfp, ctx->mass_cap, mb.K, mb.Q);
        mb.wd_stride = (long long)W * n_pops * b9k_marg_wd_table_doubles(ctx->pk.nfp, mb.K);
    } else {
        sp = make_step_plan(ctx, W, n_pops);
    }
    const B9Groups &plan = sp.plan;
    const int derive_parts = sp.derive_parts;
    // two slots (device block + pinned mirror + event) alternate, so that a block can be enqueued while its
    // predecessor is still running or waiting to be collected
    b9_ctx::McmcSlot &sl = ctx->slot[ctx->next_slot];
    if (sl.in_flight) return fail(ctx, B9_ERR_STATE, "two blocks are already outstanding: collect one with b9_mcmc_wait first");
    if (cont && (ctx->last_slot < 0 || ctx->slot[ctx->last_slot].W != W || ctx->slot[ctx->last_slot].kind != 0))
        return fail(ctx, B9_ERR_STATE, "B9_BLOCK_CONTINUE needs a previous block of this context with the same n_walkers and mode");
    const bool want_rows = blk->row_origin != nullptr;
    const size_t n_state = (size_t)W * B9_STATE_STRIDE, n_cur = (size_t)W * B9_NPARAM,
                 n_samp = (blk->samples || want_rows) ? (size_t)S * W * d : 0, n_lps = blk->lps ? (size_t)S * W : 0,
                 n_rows = want_rows ? (size_t)W * B9_ROW_LEN(d) : 0;
    // One device allocation, laid out so that the block needs ONE upload and ONE download (each small
    // pageable copy costs 10-20 us of host time, a block used to make six + four of them):
    //   [cur0][lp0][chol][origin][decided][free, ids][n_acc][state 0] | [state 1][rows][lps][samples]
    //   upload   = cur0 .. state 0        (starting state, proposal factor, moment origin, RNG streams, cleared counters)
    //   download = n_acc .. lps (.. samples when the caller wants the chain)   (acceptance count, both state parities,
    //              summary rows, log-posterior record, chain record)
    const size_t n_int = ((size_t)(d + W) + 1) / 2;                       // ints, in units of 8 bytes
    const size_t o_cur0 = 0, o_lp0 = o_cur0 + n_cur, o_chol = o_lp0 + W, o_org = o_chol + (size_t)d * d, o_dec = o_org + d,
                 o_int = o_dec + W, o_nacc = o_int + n_int, o_st0 = o_nacc + 1, o_st1 = o_st0 + n_state,
                 o_rows = o_st1 + n_state, o_lps = o_rows + n_rows, o_samp = o_lps + n_lps, n_total = o_samp + n_samp;
    const size_t up_words = o_st1, down_words = (blk->samples ? n_total : o_samp) - o_nacc;
    if (n_total * 8 > sl.cap) {
        // (a CONTINUE block reads the OTHER slot's final state, never this slot's old contents)
        if (sl.d) (void)hipFree(sl.d);
        sl.d = nullptr; sl.cap = 0;
        HIPCHK(ctx, hipMalloc(&sl.d, n_total * 8));
        sl.cap = n_total * 8;
    }
    if (n_total * 8 > sl.hcap) {
        if (sl.h) (void)hipHostFree(sl.h);
        sl.h = nullptr; sl.hcap = 0;
        HIPCHK(ctx, hipHostMalloc(&sl.h, n_total * 8, hipHostMallocMapped));     // pinned staging mirror, mapped into the device
        HIPCHK(ctx, hipHostGetDevicePointer(&sl.h_dev, sl.h, 0));
        sl.hcap = n_total * 8;
    }
    if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (!sl.rows_ready) HIPCHK(ctx, hipEventCreateWithFlags(&sl.rows_ready, hipEventDisableTiming));
    double *const dev = static_cast<double *>(sl.d), *const stage = static_cast<double *>(sl.h);
    double *d_state = dev + o_st0;                   // [2][W][stride]; the block's first launch has parity 1 and reads parity 0
    double *d_cur0 = dev + o_cur0, *d_lp0 = dev + o_lp0, *d_chol = dev + o_chol;
    unsigned long long *d_decided = reinterpret_cast<unsigned long long *>(dev + o_dec);
    int *d_free = reinterpret_cast<int *>(dev + o_int), *d_ids = d_free + d;
    unsigned long long *d_nacc = reinterpret_cast<unsigned long long *>(dev + o_nacc);
    double *d_samples = n_samp ? dev + o_samp : nullptr, *d_lps = n_lps ? dev + o_lps : nullptr;
    hipStream_t s = ctx->stream;
    {
        std::memset(stage + o_cur0, 0, (n_cur + W) * 8);
        if (!cont) {
            std::memcpy(stage + o_cur0, blk->params, n_cur * 8);
            std::memcpy(stage + o_lp0, blk->logpost, (size_t)W * 8);
        }
        std::memcpy(stage + o_chol, blk->chol, (size_t)d * d * 8);
        if (want_rows) std::memcpy(stage + o_org, blk->row_origin, (size_t)d * 8); else std::memset(stage + o_org, 0, (size_t)d * 8);
        std::memset(stage + o_dec, 0xFF, (size_t)W * 8);                 // no step published yet
        int *hi = reinterpret_cast<int *>(stage + o_int);
        std::memcpy(hi, blk->free_idx, d * sizeof(int));
        std::memcpy(hi + d, blk->walker_ids, W * sizeof(int));
        std::memset(stage + o_nacc, 0, 8);
        double *st0 = stage + o_st0;                                      // starting state -> parity 0, which K(0) (parity 1) reads
        std::memset(st0, 0, n_state * 8);
        if (!cont)
            for (int w = 0; w < W; ++w) {
                std::memcpy(st0 + (size_t)w * B9_STATE_STRIDE + B9_ST_CUR, blk->params + (size_t)w * B9_NPARAM, sizeof(double) * B9_NPARAM);
                st0[(size_t)w * B9_STATE_STRIDE + B9_ST_LP] = blk->logpost[w];
                st0[(size_t)w * B9_STATE_STRIDE + B9_ST_LPRIOR] = -INFINITY;
            }
        // one launch: the upload, read by the device from the mapped mirror, and -- continuing -- the previous block's final
        // state (stream-ordered behind its last launch) in place of the starting state
        const double *prev_final = nullptr;
        if (cont) {
            const b9_ctx::McmcSlot &pv = ctx->slot[ctx->last_slot];
            prev_final = static_cast<const double *>(pv.d) + (pv.final_parity ? pv.o_st1 : pv.o_st0);
        }
        HIPCHK(ctx, b9k_mcmc_begin(static_cast<const double *>(sl.h_dev), dev, (int)up_words, prev_final, d_cur0, d_lp0, d_state, W, s));
    }
    StepDev sd{};
    sd.d = d; sd.n_walkers = W; sd.n_pops = n_pops;
    sd.n_partial = marg ? mb.n_partial : partial_count(ctx, plan); sd.mass_cap = ctx->mass_cap; sd.heavy_parts = marg ? 0 : ctx->heavy_parts;
    sd.k0 = (unsigned)(blk->seed & 0xFFFFFFFFull); sd.k1 = (unsigned)(blk->seed >> 32);
    sd.partial_stride = marg ? mb.stride : partial_stride(ctx); sd.iso_stride = ctx->iso_stride;
    sd.state = d_state; sd.partial = marg ? mb.partial : ctx->d_partial;
    sd.cand_par = ctx->d_params; sd.cand_hdr = ctx->d_hdr; sd.cand_iso = ctx->d_iso;
    sd.chol = d_chol; sd.free_idx = d_free; sd.walker_ids = d_ids;
    sd.samples = d_samples; sd.lps = d_lps; sd.n_acc = d_nacc; sd.decided = d_decided;
    sd.rows = nullptr; sd.row_origin = dev + o_org; sd.n_steps = S;
    // (a parity's row: the hot waves' partials + one set of heavy-star partials per candidate)
    if (2 * ((long long)sd.n_partial + sd.heavy_parts) > sd.partial_stride) return fail(ctx, B9_ERR_CAPACITY, "partial buffer too small for two parities");
    {   // D0: proposal of step 0 and its isochrones -> candidate 0 of parity 1 (K(t) has parity (t + 1) & 1)
        McmcDev mc{};
        mc.enabled = 1; mc.d = d; mc.n_walkers = W; mc.has_prev = 0; mc.pin = 0; mc.row = 0;
        mc.cur = d_cur0; mc.lp_cur = d_lp0; mc.chol = d_chol; mc.free_idx = d_free; mc.walker_ids = d_ids;
        mc.k0 = sd.k0; mc.k1 = sd.k1; mc.step = (unsigned long long)blk->step0; mc.n_acc = d_nacc;
        const size_t rows = (size_t)W * n_pops, c10 = 2;     // (parity 1, candidate 0)
        HIPCHK(ctx, b9k_derive_iso(ctx->pk, sd.cand_par + c10 * W * B9_NPARAM, W, n_pops, sd.cand_hdr + c10 * rows,
                                   sd.cand_iso + c10 * rows * ctx->iso_stride, ctx->iso_stride, ctx->mass_cap,
                                   mc, ctx->pr, B9Prev{nullptr, 0, 0, nullptr, nullptr}, s));
        if (marg)       // ... and its node tables (every later candidate's are built inside k_marg_step)
            HIPCHK(ctx, b9k_marg_tables(ctx->pk, sd.cand_hdr + c10 * rows, sd.cand_iso + c10 * rows * ctx->iso_stride, ctx->iso_stride, ctx->mass_cap,
                                        sd.cand_par + c10 * W * B9_NPARAM, W, n_pops, mb.K, mb.Q, ctx->d_marg_tab + c10 * rows * mb.tab_doubles,
                                        ctx->st.n_wd > 0 ? ctx->d_marg_wd_tab + c10 * mb.wd_stride : nullptr, s));
    }
    long t_slot = -1;
    int t_covered = 0;
    for (int t = 0; t < S; ++t) {
        sd.set = (t + 1) & 1; sd.has_prev = t > 0; sd.derive_next = t + 1 < S; sd.row = t - 1;
        sd.step = (unsigned long long)(blk->step0 + t);
        // Timing: every ctx->timing-th launch opens an event bracket that spans timing_group consecutive launches of
        // this kernel (never past the block's last one), so the two event records cost 1/group of what a bracket
        // around a single launch adds; the bracket's time / its launch count is the kernel's launch period.
        if (t_slot < 0) {
            int rc = timing_begin(ctx, s, &t_slot);
            if (rc) return rc;
            t_covered = 0;
        } else if (ctx->timing > 0) ctx->launch_no++;
        if (marg)
            HIPCHK(ctx, b9k_marg_step(ctx->pk, ctx->st, sd, ctx->pr, mb.K, mb.Q, ctx->marg_prune, ctx->d_marg_tab, ctx->d_marg_wd_tab, mb.wd_stride,
                                      ctx->d_marg_shares, s));
        else
            HIPCHK(ctx, b9k_mcmc_step(ctx->pk, ctx->st, sd, ctx->pr, plan, ctx->heavy_parts, derive_parts, ctx->derive_order, s));
        if (t_slot >= 0 && (++t_covered >= ctx->timing_group || t == S - 1)) {
            ctx->ev_count[t_slot] = t_covered;
            int rc = timing_end(ctx, s, t_slot);
            if (rc) return rc;
            t_slot = -1;
        }
    }
    sd.set = (S + 1) & 1; sd.has_prev = 1; sd.derive_next = 0; sd.row = S - 1;
    sd.step = (unsigned long long)(blk->step0 + S);
    sd.rows = want_rows ? dev + o_rows : nullptr;
    // a block whose chain record stays on the device needs no download: its last launch writes what the host reads (final
    // state, accepted counts, summary rows) into the mapped mirror as well
    const bool zero_copy = !blk->samples && !blk->lps;
    double *const mirror = static_cast<double *>(sl.h_dev);
    sd.host_state = zero_copy ? mirror + (((S + 1) & 1) ? o_st1 : o_st0) : nullptr;
    sd.host_rows = (zero_copy && want_rows) ? mirror + o_rows : nullptr;
    HIPCHK(ctx, b9k_mcmc_finish(ctx->pk, sd, ctx->pr, s));
    const bool rows_event = want_rows && (blk->flags & B9_BLOCK_ROWS_EVENT) != 0;
    if (rows_event) HIPCHK(ctx, hipEventRecord(sl.rows_ready, s));
    blk->d_rows = want_rows ? (void *)(dev + o_rows) : nullptr;
    blk->rows_ready = rows_event ? (void *)sl.rows_ready : nullptr;
    if (!zero_copy) HIPCHK(ctx, hipMemcpyAsync(stage + o_nacc, dev + o_nacc, down_words * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipEventRecord(sl.done, s));
    sl.kind = 0;
    sl.W = W; sl.final_parity = (S + 1) & 1;
    sl.o_nacc = o_nacc; sl.o_st0 = o_st0; sl.o_st1 = o_st1; sl.o_samp = o_samp; sl.o_lps = o_lps; sl.n_samp = n_samp; sl.n_lps = n_lps;
    sl.o_rows = o_rows; sl.n_rows = n_rows; sl.host_samples = blk->samples != nullptr;
    sl.in_flight = true; sl.owner = blk;
    ctx->last_slot = ctx->next_slot;
    ctx->next_slot ^= 1;
    return async ? B9_OK : collect_block(ctx, sl, blk);
}

/* Device-resident Metropolis block, given-mass mode, tree-speculative launches (TreeDev in b9_device.h): `depth` steps per launch.
 * Launch sequence for S steps, M = ceil(S / depth):   B  P  K(0) K(1) ... K(M-1)  F
 *   B    = k_tree_begin: the upload from the mapped mirror; the starting state into both parities' state rows
 *   P    = k_mcmc_tree, prologue: derives the first tree (2^depth - 1 candidates) from the starting state
 *   K(m) = k_mcmc_tree: walks the tree K(m-1) evaluated (the sequential accept tests of its `depth` steps), evaluates the tree
 *          rooted at the resulting state, derives K(m+1)'s tree for every possible outcome of its own
 *   F    = k_tree_finish: the last walk, final state, summary rows.
 * Same block contract as run_block_fused (slots, mapped mirror, B9_BLOCK_ASYNC / CONTINUE, rows in HBM behind rows_ready). */
int run_block_tree(b9_ctx *ctx, b9_mcmc_block *blk, const TreePlan &tp)
{
    const int W = blk->n_walkers, d = blk->n_free, S = blk->n_steps, n_pops = ctx->opt.n_pops, depth = tp.depth;
    const bool cont = (blk->flags & B9_BLOCK_CONTINUE) != 0, async = (blk->flags & B9_BLOCK_ASYNC) != 0;
    int rc = ensure_tree_buffers(ctx, W, n_pops, tp);
    if (rc) return rc;
    b9_ctx::McmcSlot &sl = ctx->slot[ctx->next_slot];
    if (sl.in_flight) return fail(ctx, B9_ERR_STATE, "two blocks are already outstanding: collect one with b9_mcmc_wait first");
    if (cont && (ctx->last_slot < 0 || ctx->slot[ctx->last_slot].W != W || ctx->slot[ctx->last_slot].kind != 2))
        return fail(ctx, B9_ERR_STATE, "B9_BLOCK_CONTINUE needs a previous block of this context with the same n_walkers and mode");
    const bool want_rows = blk->row_origin != nullptr;
    const size_t n_state = (size_t)W * B9_TREE_STATE_STRIDE,
                 n_samp = (blk->samples || want_rows) ? (size_t)S * W * d : 0, n_lps = blk->lps ? (size_t)S * W : 0,
                 n_rows = want_rows ? (size_t)W * B9_ROW_LEN(d) : 0;
    //   [chol][origin][free, ids][state 0][state 1] | [rows][lps][samples][step table]        upload = chol .. state 1
    const size_t n_int = ((size_t)(d + W) + 1) / 2;
    const size_t tab_steps = (size_t)S + B9_TREE_MAX_DEPTH, n_tab = (size_t)W * tab_steps * B9_TREE_TAB_ROW;
    const size_t o_chol = 0, o_org = o_chol + (size_t)d * d, o_int = o_org + d, o_st0 = o_int + n_int, o_st1 = o_st0 + n_state,
                 o_rows = o_st1 + n_state, o_lps = o_rows + n_rows, o_samp = o_lps + n_lps, o_tab = o_samp + n_samp, n_total = o_tab + n_tab;
    const size_t up_words = o_rows;
    if (n_total * 8 > sl.cap) {
        if (sl.d) (void)hipFree(sl.d);
        sl.d = nullptr; sl.cap = 0;
        HIPCHK(ctx, hipMalloc(&sl.d, n_total * 8));
        sl.cap = n_total * 8;
    }
    if (n_total * 8 > sl.hcap) {
        if (sl.h) (void)hipHostFree(sl.h);
        sl.h = nullptr; sl.hcap = 0;
        HIPCHK(ctx, hipHostMalloc(&sl.h, n_total * 8, hipHostMallocMapped));
        HIPCHK(ctx, hipHostGetDevicePointer(&sl.h_dev, sl.h, 0));
        sl.hcap = n_total * 8;
    }
    if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (!sl.rows_ready) HIPCHK(ctx, hipEventCreateWithFlags(&sl.rows_ready, hipEventDisableTiming));
    double *const dev = static_cast<double *>(sl.d), *const stage = static_cast<double *>(sl.h);
    hipStream_t s = ctx->stream;
    {
        std::memcpy(stage + o_chol, blk->chol, (size_t)d * d * 8);
        if (want_rows) std::memcpy(stage + o_org, blk->row_origin, (size_t)d * 8); else std::memset(stage + o_org, 0, (size_t)d * 8);
        int *hi = reinterpret_cast<int *>(stage + o_int);
        std::memcpy(hi, blk->free_idx, d * sizeof(int));
        std::memcpy(hi + d, blk->walker_ids, W * sizeof(int));
        std::memset(stage + o_st0, 0, 2 * n_state * 8);
        if (!cont)
            for (int p = 0; p < 2; ++p)
                for (int w = 0; w < W; ++w) {
                    double *row = stage + (p ? o_st1 : o_st0) + (size_t)w * B9_TREE_STATE_STRIDE;
                    std::memcpy(row + B9_TS_CUR, blk->params + (size_t)w * B9_NPARAM, sizeof(double) * B9_NPARAM);
                    row[B9_TS_LP] = blk->logpost[w];
                }
        const double *prev_final = nullptr;
        if (cont) {
            const b9_ctx::McmcSlot &pv = ctx->slot[ctx->last_slot];
            prev_final = static_cast<const double *>(pv.d) + (pv.final_parity ? pv.o_st1 : pv.o_st0);
        }
        HIPCHK(ctx, b9k_tree_begin(static_cast<const double *>(sl.h_dev), dev, (int)up_words, prev_final, dev + o_st0, W, s));
    }
    TreeDev td{};
    td.d = d; td.n_walkers = W; td.n_pops = n_pops; td.depth = depth;
    td.n_groups = tp.n_groups; td.heavy_parts = ctx->heavy_parts; td.mass_cap = ctx->mass_cap;
    td.part_stride = (int)(((size_t)tp.n_groups * 4 + ctx->heavy_parts + 1) & ~(size_t)1);
    td.k0 = (unsigned)(blk->seed & 0xFFFFFFFFull); td.k1 = (unsigned)(blk->seed >> 32);
    td.iso_stride = ctx->iso_stride;
    td.state = dev + o_st0; td.partial = ctx->d_tree_partial;
    td.cand_par = ctx->d_tree_par; td.cand_hdr = ctx->d_tree_hdr; td.cand_iso = ctx->d_tree_iso;
    td.chol = dev + o_chol; td.free_idx = reinterpret_cast<int *>(dev + o_int); td.walker_ids = td.free_idx + d;
    td.samples = n_samp ? dev + o_samp : nullptr; td.lps = n_lps ? dev + o_lps : nullptr;
    td.row_origin = dev + o_org; td.n_steps = S;
    td.step_tab = dev + o_tab; td.tab_steps = (int)tab_steps; td.block_step0 = (unsigned long long)blk->step0;
    const int tiles_arg = tp.group_tiles;
    const int M = (S + depth - 1) / depth;
    {   // P: the block's first tree from the starting state -> candidates of parity 0, outcome slot 0
        td.set = 1; td.levels_prev = 0; td.levels = 0; td.derive_mode = 2; td.row = 0;
        td.step = (unsigned long long)blk->step0; td.next_step = (unsigned long long)blk->step0;
        HIPCHK(ctx, b9k_mcmc_tree(ctx->pk, ctx->st, td, ctx->pr, tiles_arg, tp.derive_parts, s));
    }
    long t_slot = -1;
    int t_covered = 0;
    for (int m = 0; m < M; ++m) {
        td.set = m & 1;
        td.levels_prev = m > 0 ? depth : 0;
        td.levels = std::min(depth, S - m * depth);
        td.derive_mode = (m + 1 < M) ? 1 : 0;
        td.row = (m - 1) * depth;
        td.step = (unsigned long long)(blk->step0 + (long long)m * depth);
        td.next_step = td.step + (unsigned)depth;
        if (t_slot < 0) {
            rc = timing_begin(ctx, s, &t_slot);
            if (rc) return rc;
            t_covered = 0;
        } else if (ctx->timing > 0) ctx->launch_no++;
        HIPCHK(ctx, b9k_mcmc_tree(ctx->pk, ctx->st, td, ctx->pr, tiles_arg, tp.derive_parts, s));
        if (t_slot >= 0 && (++t_covered >= ctx->timing_group || m == M - 1)) {
            ctx->ev_count[t_slot] = t_covered;
            rc = timing_end(ctx, s, t_slot);
            if (rc) return rc;
            t_slot = -1;
        }
    }
    {   // F
        td.set = M & 1;
        td.levels_prev = std::min(depth, S - (M - 1) * depth);
        td.levels = 0; td.derive_mode = 0;
        td.row = (M - 1) * depth;
        td.step = (unsigned long long)(blk->step0 + S); td.next_step = td.step;
        td.rows = want_rows ? dev + o_rows : nullptr;
        const bool zero_copy = !blk->samples && !blk->lps;
        double *const mirror = static_cast<double *>(sl.h_dev);
        td.host_state = zero_copy ? mirror + ((M & 1) ? o_st1 : o_st0) : nullptr;
        td.host_rows = (zero_copy && want_rows) ? mirror + o_rows : nullptr;
        HIPCHK(ctx, b9k_tree_finish(td, ctx->pr, s));
        const bool rows_event = want_rows && (blk->flags & B9_BLOCK_ROWS_EVENT) != 0;
        if (rows_event) HIPCHK(ctx, hipEventRecord(sl.rows_ready, s));
        blk->d_rows = want_rows ? (void *)(dev + o_rows) : nullptr;
        blk->rows_ready = rows_event ? (void *)sl.rows_ready : nullptr;
        if (!zero_copy) HIPCHK(ctx, hipMemcpyAsync(stage + o_st0, dev + o_st0, ((blk->samples ? o_tab : o_samp) - o_st0) * 8, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(ctx, hipEventRecord(sl.done, s));
    sl.kind = 2; sl.W = W; sl.final_parity = M & 1;
    sl.o_st0 = o_st0; sl.o_st1 = o_st1; sl.o_samp = o_samp; sl.o_lps = o_lps; sl.n_samp = n_samp; sl.n_lps = n_lps;
    sl.o_rows = o_rows; sl.n_rows = n_rows; sl.host_samples = blk->samples != nullptr;
    sl.in_flight = true; sl.owner = blk;
    ctx->last_slot = ctx->next_slot;
    ctx->next_slot ^= 1;
    return async ? B9_OK : collect_block(ctx, sl, blk);
}

/* Device-resident Metropolis block with TWO launches per step (marginalised mode; b9_tuning.two_launch_steps):
 *   D(0) L(0)  D(1) L(1)  ...  D(S-1) L(S-1)  F  [R]
 *   D(t) = k_derive_iso: finishes step t-1 (sum + prior + accept; t > 0), proposes step t, derives its isochrones
 *   L(t) = the star likelihood of step t's proposals;   F = k_finalize: finishes the last step;
 *   R    = k_chain_rows: the block's per-walker summary rows, condensed from the chain record on the device.
 * Same contract as the fused path: one pinned mirror per slot for the upload and the download, B9_BLOCK_ASYNC /
 * B9_BLOCK_CONTINUE / summary rows in HBM behind rows_ready -- a star launch here takes milliseconds, so none of this is for
 * speed; it gives a multi-GPU driver ONE way to run blocks and to read rows, whatever the evaluation mode. */
int run_block_two_launch(b9_ctx *ctx, b9_mcmc_block *blk, const B9Groups &plan)
{
    const int W = blk->n_walkers, d = blk->n_free, S = blk->n_steps, n_pops = ctx->opt.n_pops;
    const bool cont = (blk->flags & B9_BLOCK_CONTINUE) != 0, async = (blk->flags & B9_BLOCK_ASYNC) != 0;
    b9_ctx::McmcSlot &sl = ctx->slot[ctx->next_slot];
    if (sl.in_flight) return fail(ctx, B9_ERR_STATE, "two blocks are already outstanding: collect one with b9_mcmc_wait first");
    if (cont && (ctx->last_slot < 0 || ctx->slot[ctx->last_slot].W != W || ctx->slot[ctx->last_slot].kind != 1))
        return fail(ctx, B9_ERR_STATE, "B9_BLOCK_CONTINUE needs a previous block of this context with the same n_walkers and mode");
    const bool want_rows = blk->row_origin != nullptr;
    const size_t n_cur = (size_t)W * B9_NPARAM, n_samp = (blk->samples || want_rows) ? (size_t)S * W * d : 0,
                 n_lps = blk->lps ? (size_t)S * W : 0, n_rows = want_rows ? (size_t)W * B9_ROW_LEN(d) : 0;
    // [chol][origin][free, ids][n_acc][cur: two halves][lp: two halves][rows][lps][samples]
    //  upload = chol .. first half of lp's start state;  download = n_acc .. lps (.. samples when the caller wants the chain)
    const size_t n_int = ((size_t)(d + W) + 1) / 2;
    const size_t o_chol = 0, o_org = o_chol + (size_t)d * d, o_int = o_org + d, o_nacc = o_int + n_int, o_cur = o_nacc + 1,
                 o_lp = o_cur + 2 * n_cur, o_rows = o_lp + 2 * (size_t)W, o_lps = o_rows + n_rows, o_samp = o_lps + n_lps,
                 n_total = o_samp + n_samp;
    if (n_total * 8 > sl.cap) {
        if (sl.d) (void)hipFree(sl.d);
        sl.d = nullptr; sl.cap = 0;
        HIPCHK(ctx, hipMalloc(&sl.d, n_total * 8));
        sl.cap = n_total * 8;
    }
    if (n_total * 8 > sl.hcap) {
        if (sl.h) (void)hipHostFree(sl.h);
        sl.h = nullptr; sl.hcap = 0;
        HIPCHK(ctx, hipHostMalloc(&sl.h, n_total * 8, hipHostMallocMapped));
        HIPCHK(ctx, hipHostGetDevicePointer(&sl.h_dev, sl.h, 0));
        sl.hcap = n_total * 8;
    }
    if (!sl.done) HIPCHK(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (!sl.rows_ready) HIPCHK(ctx, hipEventCreateWithFlags(&sl.rows_ready, hipEventDisableTiming));
    double *const dev = static_cast<double *>(sl.d), *const stage = static_cast<double *>(sl.h);
    hipStream_t s = ctx->stream;
    // upload: proposal factor, moment origin, RNG streams, cleared counter and (unless continuing) the starting state
    std::memcpy(stage + o_chol, blk->chol, (size_t)d * d * 8);
    if (want_rows) std::memcpy(stage + o_org, blk->row_origin, (size_t)d * 8); else std::memset(stage + o_org, 0, (size_t)d * 8);
    int *hi = reinterpret_cast<int *>(stage + o_int);
    std::memcpy(hi, blk->free_idx, d * sizeof(int));
    std::memcpy(hi + d, blk->walker_ids, W * sizeof(int));
    std::memset(stage + o_nacc, 0, 8);
    HIPCHK(ctx, hipMemcpyAsync(dev + o_chol, stage + o_chol, (o_cur - o_chol) * 8, hipMemcpyHostToDevice, s));
    if (!cont) {
        std::memcpy(stage + o_cur, blk->params, n_cur * 8);
        std::memcpy(stage + o_lp, blk->logpost, (size_t)W * 8);
        HIPCHK(ctx, hipMemcpyAsync(dev + o_cur, stage + o_cur, n_cur * 8, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(dev + o_lp, stage + o_lp, (size_t)W * 8, hipMemcpyHostToDevice, s));
    } else {      // the previous block's final half (stream-ordered behind its last launch) -> this block's half 0
        const b9_ctx::McmcSlot &pv = ctx->slot[ctx->last_slot];
        const double *pd = static_cast<const double *>(pv.d);
        HIPCHK(ctx, hipMemcpyAsync(dev + o_cur, pd + pv.o_cur, n_cur * 8, hipMemcpyDeviceToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(dev + o_lp, pd + pv.o_lp, (size_t)W * 8, hipMemcpyDeviceToDevice, s));
    }
    McmcDev mc{};
    mc.enabled = 1; mc.d = d; mc.n_walkers = W;
    mc.cur = dev + o_cur; mc.lp_cur = dev + o_lp;
    mc.chol = dev + o_chol;
    mc.free_idx = reinterpret_cast<int *>(dev + o_int); mc.walker_ids = mc.free_idx + d;
    mc.samples = n_samp ? dev + o_samp : nullptr; mc.lps = n_lps ? dev + o_lps : nullptr;
    mc.n_acc = reinterpret_cast<unsigned long long *>(dev + o_nacc);
    mc.k0 = (unsigned)(blk->seed & 0xFFFFFFFFull); mc.k1 = (unsigned)(blk->seed >> 32);
    const int n_part = partial_count(ctx, plan);
    for (int t = 0; t < S; ++t) {
        const Bufs bf = buffer_set(ctx, t & 1), bp = buffer_set(ctx, (t & 1) ^ 1);
        mc.step = (unsigned long long)(blk->step0 + t);     // the step being proposed
        mc.has_prev = t > 0;
        mc.pin = t > 0 ? (t - 1) & 1 : 0;                   // state half on entry
        mc.row = t - 1;                                     // chain row of the step being finished
        const B9Prev prev{ctx->d_partial, n_part, partial_stride(ctx), bp.hdr, bp.params};
        HIPCHK(ctx, b9k_derive_iso(ctx->pk, bf.params, W, n_pops, bf.hdr, bf.iso, ctx->iso_stride, ctx->mass_cap,
                                   mc, ctx->pr, prev, s));
        const int rc = launch_stars(ctx, bf, W, nullptr, plan, s);
        if (rc) return rc;
    }
    {   // finish the last step
        const Bufs bf = buffer_set(ctx, (S - 1) & 1);
        mc.step = (unsigned long long)(blk->step0 + S - 1);
        mc.has_prev = 0;
        mc.pin = S > 1 ? (S - 2) & 1 : 0;                   // the half D(S-1) wrote (or the initial half)
        if (S > 1) mc.pin ^= 1;
        mc.row = S - 1;
        HIPCHK(ctx, b9k_finalize(bf.hdr, ctx->d_partial, n_part, partial_stride(ctx), n_pops, bf.params, ctx->pr, W,
                                 ctx->d_logpost, nullptr, ctx->st.n, mc, s));
    }
    const int fin = mc.pin ^ 1;                             // half that holds the final state
    const size_t o_cur_fin = o_cur + (size_t)fin * n_cur, o_lp_fin = o_lp + (size_t)fin * W;
    if (want_rows) {
        StepDev sd{};
        sd.d = d; sd.n_walkers = W; sd.n_steps = S; sd.samples = mc.samples; sd.free_idx = mc.free_idx;
        sd.row_origin = dev + o_org; sd.rows = dev + o_rows; sd.host_rows = nullptr;
        HIPCHK(ctx, b9k_chain_rows(sd, dev + o_cur_fin, dev + o_lp_fin, s));
    }
    const bool rows_event = want_rows && (blk->flags & B9_BLOCK_ROWS_EVENT) != 0;
    if (rows_event) HIPCHK(ctx, hipEventRecord(sl.rows_ready, s));
    blk->d_rows = want_rows ? (void *)(dev + o_rows) : nullptr;
    blk->rows_ready = rows_event ? (void *)sl.rows_ready : nullptr;
    const size_t down_end = blk->samples ? n_total : o_samp;
    HIPCHK(ctx, hipMemcpyAsync(stage + o_nacc, dev + o_nacc, (down_end - o_nacc) * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipEventRecord(sl.done, s));
    sl.kind = 1; sl.W = W; sl.final_parity = fin;
    sl.o_nacc = o_nacc; sl.o_cur = o_cur_fin; sl.o_lp = o_lp_fin; sl.o_samp = o_samp; sl.o_lps = o_lps; sl.o_rows = o_rows;
    sl.n_samp = n_samp; sl.n_lps = n_lps; sl.n_rows = n_rows; sl.host_samples = blk->samples != nullptr;
    sl.in_flight = true; sl.owner = blk;
    ctx->last_slot = ctx->next_slot;
    ctx->next_slot ^= 1;
    return async ? B9_OK : collect_block(ctx, sl, blk);
}


}  // namespace

extern "C" {

/* Device-resident Metropolis block (the caller of the hot path).  Given-mass mode
 * runs the fused one-launch step (run_block_fused above); what follows is the two-launch step of the
 * marginalised mode.
 * Launch sequence for S steps:  D(0) L(0)  D(1) L(1)  ...  D(S-1) L(S-1)  F
 *   D(t) = k_derive_iso: finishes step t-1 (sum + prior + accept; t > 0), proposes step t, derives
 *   L(t) = star likelihood of step t's proposals;   F = k_finalize: finishes the last step.
 * Two launches per step; buffers and walker state ping-pong between two halves. */
int b9_mcmc_run_block(b9_ctx *ctx, b9_mcmc_block *blk)
{
    if (!ctx || !blk || blk->n_walkers < 1 || blk->n_steps < 0 || blk->n_free < 1 || blk->n_free > 11 ||
        !blk->free_idx || !blk->chol || !blk->walker_ids || !blk->params || !blk->logpost)
        return B9_ERR_INVALID;
    int rc = check_ready(ctx);
    if (rc) return rc;
    const int W = blk->n_walkers, d = blk->n_free, S = blk->n_steps, n_pops = ctx->opt.n_pops;
    for (int i = 0; i < d; ++i)
        if (blk->free_idx[i] < 0 || blk->free_idx[i] >= B9_NPARAM) return fail(ctx, B9_ERR_INVALID, "free_idx out of range");
    if (S == 0) { blk->n_accept = 0; return B9_OK; }
    // (the work buffers are sized for the walker count: they must not be re-allocated under an enqueued block)
    for (const auto &sl : ctx->slot)
        if (sl.in_flight && sl.W != W) return fail(ctx, B9_ERR_STATE, "collect the outstanding block(s) before running a block with another number of walkers");
    const B9Groups plan = make_plan(ctx, W, n_pops);
    rc = ensure_capacity(ctx, W, n_pops, (size_t)partial_stride(ctx) * W, false);
    if (rc) return rc;
    if (ctx->opt.mode == B9_MODE_GIVEN_MASS && !ctx->two_launch_steps) {
        const TreePlan tp = make_tree_plan(ctx, W, n_pops);
        return tp.depth >= 2 ? run_block_tree(ctx, blk, tp) : run_block_fused(ctx, blk, false);
    }
    if (ctx->opt.mode == B9_MODE_MARGINALISED && !ctx->two_launch_steps && marg_fused_ok(ctx)) return run_block_fused(ctx, blk, true);
    return run_block_two_launch(ctx, blk, plan);
}

int b9_mcmc_wait(b9_ctx *ctx, b9_mcmc_block *blk)
{
    if (!ctx || !blk || !blk->params || !blk->logpost) return B9_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // blocks are collected in the order they were enqueued: the older outstanding one is in next_slot when both
    // are in flight, else in the other slot
    for (int k = 0; k < 2; ++k) {
        b9_ctx::McmcSlot &sl = ctx->slot[(ctx->next_slot + k) & 1];
        if (sl.in_flight) {
            if (sl.owner != blk) return fail(ctx, B9_ERR_STATE, "b9_mcmc_wait: blocks must be collected in the order they were enqueued");
            return collect_block(ctx, sl, blk);
        }
    }
    return fail(ctx, B9_ERR_STATE, "b9_mcmc_wait: no block is outstanding");
}

}  // extern "C"
