// b9_capi_margplan.cpp -- the marginalised mode's CATALOGUE PLAN: in which order the star chunks are dispatched and, on small
// catalogues, how many workgroups share each chunk's window -- from the cost the kernel itself measures at a reference row.
//
// A 64-star chunk costs what the union of its stars' node windows holds; a chunk of giants walks ten times the units of a
// chunk of dwarfs, and photometric spread (round 4's proxy) orders chunks but does not measure them.  So once per
// (catalogue, pack, priors, options) the unsplit star kernel runs on ONE row -- the prior means, clamped into the grid --
// with every wave counting the (16 nodes x one mass ratio) units it evaluates (the COST instance; b9k_star_marg_cost), and
//   * the dispatch order becomes descending measured cost;
//   * a catalogue below 512 chunk-populations gives chunk c  n_c = ceil(cost_c / P) pieces (1 .. 32), P = the piece size that
//     fills the chip's workgroup slots once (never under b9_tuning.marg_piece_units units): pieces cost about the same, the launch no
//     longer lasts as long as its heaviest chunk (round 4 split every chunk 8 ways: 10k stars, one chain: the heaviest
//     piece lived 27 us, the median 6.5).
// WHICH stars share a chunk stays what b9_load_stars made it -- neighbours along the first principal component of the
// magnitudes.  Round 5 measured the alternatives with this same counting pass (wave-units on the reference row, 50k stars x 8
// filters, 4 x 4 grid): that order 8012; stars re-dealt by their best node on the reference table 12812, by (best mass ratio,
// best node) 8868 (the argmax wanders along the mass / mass-ratio degeneracy: photometric twins end up chunks apart); blocks
// of 2 / 4 / 8 / 16 chunks re-sorted by the second component 8091 / 8371 / 9101 / 10467.  A star's own halo of live terms
// (80 of the 163 a wave evaluates), not the union over its neighbours, is what a wave pays for.
// Everything here is a function of the catalogue, the pack, the priors and the options -- the same on every rank, whatever
// walkers it holds -- because the pieces decide how a star's sum rounds and a walker's chain must not depend on its
// neighbours.  Speed only otherwise: any plan gives a correct sum.
#include "b9_ctx.h"
#include <cstdint>

using namespace b9i;

namespace {

constexpr int kMaxPieces = 32;

int upload_ints(b9_ctx *ctx, const std::vector<int> &v, const int **out)
{
    void *d = nullptr;
    HIPCHK(ctx, hipMalloc(&d, sizeof(int) * std::max<size_t>(v.size(), 1)));
    ctx->marg_plan_allocs.push_back(d);
    if (!v.empty()) HIPCHK(ctx, hipMemcpy(d, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
    *out = static_cast<const int *>(d);
    return B9_OK;
}

// the reference row: the prior means, the three grid coordinates clamped into the pack's axes
void reference_row(const b9_ctx *ctx, double *row)
{
    const DevPriors &pr = ctx->pr;
    for (int k = 0; k < B9_NPARAM; ++k) row[k] = std::isfinite(pr.mean[k]) ? pr.mean[k] : 0.0;
    auto clamp_axis = [](double v, const std::vector<double> &ax) {
        if (ax.empty()) return v;
        const double lo = ax.front(), hi = ax.back(), eps = 1e-9 * std::max(1.0, std::fabs(hi - lo));
        if (!(v >= lo && v <= hi)) v = 0.5 * (lo + hi);
        return std::min(std::max(v, lo + eps), hi - eps);
    };
    row[B9_P_LOGAGE] = clamp_axis(row[B9_P_LOGAGE], ctx->h_log_age);
    row[B9_P_FEH] = clamp_axis(row[B9_P_FEH], ctx->h_feh);
    if (ctx->h_y.size() > 1) { row[B9_P_Y] = clamp_axis(row[B9_P_Y], ctx->h_y); row[B9_P_Y2] = clamp_axis(row[B9_P_Y2], ctx->h_y); }
    else if (!ctx->h_y.empty()) { row[B9_P_Y] = ctx->h_y[0]; row[B9_P_Y2] = ctx->h_y[0]; }
    if (!(row[B9_P_ABS] >= 0.0)) row[B9_P_ABS] = 0.0;
    if (!(row[B9_P_LAMBDA] > 0.0 && row[B9_P_LAMBDA] < 1.0)) row[B9_P_LAMBDA] = 0.5;
}

// the counting pass on the reference row: units evaluated per (star chunk, wave)
int counting_pass(b9_ctx *ctx, const double *row, int K, int Q, std::vector<unsigned> &h_cost, bool &valid)
{
    const int n_pops = ctx->opt.n_pops, n_mc = ctx->st.mg_pad / 64;
    const Bufs bf = buffer_set(ctx, 0);
    unsigned *d_cost = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_cost, sizeof(unsigned) * 4 * n_mc));
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemsetAsync(d_cost, 0, sizeof(unsigned) * 4 * n_mc, s);
    if (e == hipSuccess) e = b9k_derive_iso_rows(ctx->pk, row, bf.params, 1, n_pops, bf.hdr, bf.iso, ctx->iso_stride, ctx->mass_cap, s);
    if (e == hipSuccess) e = b9k_marg_tables(ctx->pk, bf.hdr, bf.iso, ctx->iso_stride, ctx->mass_cap, bf.params, 1, n_pops, K, Q, ctx->d_marg_tab, nullptr, s);
    if (e == hipSuccess) e = b9k_star_marg_cost(ctx->pk, ctx->st, bf.hdr, ctx->mass_cap, bf.params, n_pops, ctx->d_partial, partial_stride(ctx), K, Q,
                                                ctx->marg_prune, ctx->d_marg_tab, d_cost, s);
    h_cost.assign(4 * (size_t)n_mc, 0u);
    std::vector<IsoHdr> h_hdr(n_pops);
    if (e == hipSuccess) e = hipMemcpyAsync(h_cost.data(), d_cost, sizeof(unsigned) * h_cost.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(h_hdr.data(), bf.hdr, sizeof(IsoHdr) * n_pops, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_cost);
    if (e != hipSuccess) return fail(ctx, B9_ERR_HIP, std::string("marginalised catalogue plan: ") + hipGetErrorString(e));
    valid = true;
    for (int k = 0; k < n_pops; ++k) valid = valid && h_hdr[k].valid;
    return B9_OK;
}

}  // namespace

namespace b9i {

int ensure_marg_plan(b9_ctx *ctx)
{
    if (ctx->opt.mode != B9_MODE_MARGINALISED || ctx->marg_plan_ok) return B9_OK;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    const int n_pops = ctx->opt.n_pops, n_mc = ctx->st.mg_pad / 64;
    const int K = ctx->opt.marg_iso_increm > 0 ? ctx->opt.marg_iso_increm : 1, Q = ctx->opt.marg_n_q > 0 ? ctx->opt.marg_n_q : 1;
    free_all(ctx->marg_plan_allocs);
    // back to the load-time order (until the plan's own is in place)
    ctx->st.mg_n_pieces = 0; ctx->st.mg_piece = nullptr; ctx->st.mg_share_base = nullptr;
    ctx->st.marg_order = ctx->marg_order_spread;
    std::vector<double> cost(n_mc, -1.0);
    bool measured = false;
    {
        int rc = ensure_capacity(ctx, 1, n_pops, (size_t)partial_stride(ctx), false);
        if (rc) return rc;
        rc = ensure_marg_table(ctx, 1, n_pops, K, Q);
        if (rc) return rc;
        double row[B9_NPARAM];
        reference_row(ctx, row);
        std::vector<unsigned> h_cost;
        if ((rc = counting_pass(ctx, row, K, Q, h_cost, measured))) return rc;
        if (ctx->plan_debug && measured) { double t = 0.0; for (unsigned c : h_cost) t += c; fprintf(stderr, "[marg plan] %.0f wave-units on the reference row\n", t); }
        // a chunk's cost: its busiest wave's units (the four waves share the chunk's life), plus the chunk's fixed work in the
        // same currency (entry, level-1 boxes, merge: about two units)
        if (measured)
            for (int c = 0; c < n_mc; ++c)
                cost[c] = 2.0 + (double)std::max(std::max(h_cost[4 * c], h_cost[4 * c + 1]), std::max(h_cost[4 * c + 2], h_cost[4 * c + 3]));
    }
    // ---- dispatch order: descending measured cost (ties: chunk index).  The reference row's isochrone did not exist (no two
    // common EEPs): the load-time order, and every chunk counts the same
    std::vector<int> order(n_mc);
    int rc = B9_OK;
    if (measured) {
        ctx->marg_cost = cost;
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
        if ((rc = upload_ints(ctx, order, &ctx->st.marg_order))) return rc;
    } else {
        HIPCHK(ctx, hipMemcpy(order.data(), ctx->marg_order_spread, sizeof(int) * n_mc, hipMemcpyDeviceToHost));
        std::fill(cost.begin(), cost.end(), 1.0);
    }
    // ---- pieces
    if (b9k_marg_split(n_mc, n_pops)) {
        const int waves = ctx->pk.nfp >= 16 ? 4 : (n_pops == 2 ? 6 : 7);             // B9_MARG_WAVES of the sampler's instances
        const double slots = (double)ctx->n_cu * waves;
        double total = 0.0;
        for (double c : cost) total += c;
        // the smallest piece worth a workgroup (b9_tuning.marg_piece_units; measured, us per step of one chain at 4 x 4, 10k / 20k
        // stars: 2 units 32.9 / 39.0, 3: 33.1 / 37.5, 4: 31.6 / 36.4, 6: 39.1 / 51.4, 10: 49.2 / 53.5; 8 walkers: 4: 55.8 / 92.9,
        // 10: 56.3 / 79.2 -- a lone wave is latency-bound, one scalar-load round trip per table row: more resident waves hide it)
        // (the 8 x 8 grid, same shapes: 2 units 48.6 / 62.2, 4: 43.0 / 50.6, 6: 41.1 / 47.0, 8: 47.6 / 53.6, 12: 53.2 / 66.5 -- six from eight mass ratios on)
        const double piece_min = ctx->marg_piece_units > 0 ? (double)ctx->marg_piece_units : (Q >= 8 ? 6.0 : 4.0);
        const double P = measured ? std::max(total / slots, piece_min) : 1.0 / 8.0;      // (unmeasured: 8 pieces each, round 4's split)
        std::vector<int> n_piece(n_mc), base(n_mc + 1, 0), pieces;
        for (int c = 0; c < n_mc; ++c) n_piece[c] = std::min(kMaxPieces, std::max(1, (int)std::ceil(cost[c] / P)));
        for (int c = 0; c < n_mc; ++c) base[c + 1] = base[c] + n_piece[c];
        // dispatch order of the pieces: descending piece cost; a chunk's pieces stay together
        std::vector<int> by_piece(order);
        std::stable_sort(by_piece.begin(), by_piece.end(), [&](int a, int b) { return cost[a] / n_piece[a] > cost[b] / n_piece[b]; });
        for (int c : by_piece)
            for (int k = 0; k < n_piece[c]; ++k) pieces.push_back(c | (k << 20) | (n_piece[c] << 25));
        ctx->st.mg_n_pieces = (int)pieces.size();
        if (ctx->plan_debug) {
            std::vector<double> cs(cost); std::sort(cs.begin(), cs.end());
            fprintf(stderr, "[marg plan] %d chunks, cost min %.0f p50 %.0f p90 %.0f max %.0f total %.0f; P %.2f; %d pieces (max per chunk %d)\n", n_mc, cs.front(), cs[cs.size() / 2],
                    cs[cs.size() * 9 / 10], cs.back(), total, P, (int)pieces.size(), *std::max_element(n_piece.begin(), n_piece.end()));
        }
        if (ctx->plan_debug > 1)
            for (size_t k = 0; k < pieces.size(); ++k)
                fprintf(stderr, "[marg plan] position %zu: chunk %d piece %d of %d, chunk cost %.0f\n", k, pieces[k] & 0xFFFFF, (pieces[k] >> 20) & 31, (pieces[k] >> 25) & 63, cost[pieces[k] & 0xFFFFF]);
        if ((rc = upload_ints(ctx, pieces, &ctx->st.mg_piece))) return rc;
        if ((rc = upload_ints(ctx, base, &ctx->st.mg_share_base))) return rc;
    }
    ctx->marg_plan_ok = true;
    return B9_OK;
}

}  // namespace b9i
