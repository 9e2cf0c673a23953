// b9_capi_plan.cpp -- launch plans: the catalogue's canonical tile groups, b9_logpost's star launch, the fused sampler
// step and the tree-speculative step; b9_tuning -> plan fields.
#include "b9_ctx.h"

namespace b9i {

// Workgroups of the fused step resident at once for the loaded pack and options (occupancy query of that instantiation, cached)
int step_slots(b9_ctx *ctx, int n_pops)
{
    const int key = (ctx->pk.nfp * 4 + n_pops) * 65536 + ctx->mass_cap;
    if (ctx->step_occ_key != key) {
        int per_cu = 0;
        if (b9k_mcmc_step_occupancy(ctx->pk, n_pops, ctx->mass_cap, &per_cu) != hipSuccess || per_cu < 1) per_cu = 1;
        ctx->step_blocks_per_cu = per_cu;
        ctx->step_occ_key = key;
    }
    return ctx->n_cu * std::max(1, ctx->step_blocks_per_cu);
}

// The catalogue's CANONICAL tile groups (b9_star_like.hip.h): n_groups groups of group_tiles tiles, one partial sum per
// group and wave -- the grouping fixes how a walker's log-posterior ROUNDS, so it must not depend on how many walkers share
// the GPU (a chain is then the same bits on 1, 2, 4 or 8 ranks).  It is a function of the catalogue, the pack, the options
// and the device only: the tiles per workgroup the fused step wants on the REFERENCE shape of 8 walkers per GPU -- the
// smallest value that lets the hot workgroups fill <= hot_fill of one occupancy round.  b9_tuning.tiles_per_block pins it.
// A launch plan's only freedom is how many whole groups a workgroup takes.
constexpr int kReferenceWalkers = 8;

// Share of the resident-workgroup slots the hot workgroups may take.  One population: 0.7 (measured on 50k stars x 8 filters x
// 8 walkers: 3 tiles per workgroup 21.8 us per step, 1 tile 26.0, 4 tiles 25.6).  Two populations: 0.85 -- a hot workgroup
// covers half a tile per population, so the launch has twice the wave-evaluations to place and the fewer tiles per wave are
// worth the crowding (30k stars x 2 populations x 8 walkers: 3 tiles 19.7 us per step, 4 tiles 20.8) now that the derivation
// runs ahead of the decision and gets by with two parts.
static double hot_fill(int n_pops) { return n_pops == 2 ? 0.85 : 0.7; }

Groups make_groups(b9_ctx *ctx, int n_pops)
{
    const int n_tiles = ctx->st.n_pad / 256;
    int g = ctx->tiles_per_block;
    if (g <= 0) {
        const int slots = step_slots(ctx, n_pops);
        g = 1;
        // (two populations: a hot workgroup covers half of every tile of its groups, once per population -- n_pops per block)
        while (g < 8 && (long long)((n_tiles + g - 1) / g) * n_pops * kReferenceWalkers > (long long)(hot_fill(n_pops) * slots)) ++g;
    }
    g = std::max(1, std::min(g, std::max(1, n_tiles)));
    return Groups{g, (n_tiles + g - 1) / g};
}

B9Groups with_groups_per_block(const Groups &gr, int m)
{
    m = std::max(1, std::min(m, gr.n_groups));
    return B9Groups{gr.group_tiles, gr.n_groups, m, (gr.n_groups + m - 1) / m};
}

// b9_logpost's star launch (k_star_like): one group per workgroup until there are more workgroups than ~8 per CU; beyond
// that amortise the per-workgroup mass-column staging over several groups
B9Groups make_plan(b9_ctx *ctx, int n_walkers, int n_pops)
{
    const Groups gr = make_groups(ctx, n_pops);
    const long long tiles_wanted = std::max<long long>(1, std::min<long long>(8, (long long)(ctx->st.n_pad / 256) * n_walkers / 4096));
    return with_groups_per_block(gr, (int)(tiles_wanted / gr.group_tiles));
}

// Launch plan of the fused sampler step.  The launch has three kinds of workgroups (heavy-star, candidate
// derivation, hot); it is fastest when ALL of them are resident at once -- one occupancy round, every
// workgroup takes the previous step's decision exactly once -- so a workgroup takes the smallest number of canonical
// groups that lets the hot workgroups fill <= 70 % of the slots (never more than ~8 tiles: with more walkers than one
// round can hold that stays best -- every workgroup pays the decision prologue once; measured 44.6 vs 60.2 us at 32
// walkers, 83.3 vs 91.4 at 64), and the derivation is cut into as many parts as the remaining slots allow.

StepPlan make_step_plan(b9_ctx *ctx, int n_walkers, int n_pops)
{
    StepPlan sp;
    const Groups gr = make_groups(ctx, n_pops);
    const int slots = step_slots(ctx, n_pops);
    const int key = (ctx->pk.nfp * 4 + n_pops) * 65536 + ctx->mass_cap;
    const int full_parts = (ctx->mass_cap * (ctx->pk.nfp + 1) + 255) / 256;
    const int m_max = std::max(1, 8 / gr.group_tiles);
    int m = 1;
    while (m < m_max && (long long)((gr.n_groups + m - 1) / m) * n_pops * n_walkers > (long long)(hot_fill(n_pops) * slots)) ++m;
    sp.plan = with_groups_per_block(gr, m);
    int parts = ctx->derive_parts;
    if (parts <= 0) {
        const long long room = (long long)(0.9 * slots) - (long long)sp.plan.n_blocks * n_pops * n_walkers - (long long)n_walkers * ctx->heavy_parts;
        parts = (int)std::max<long long>(2, room / ((long long)n_walkers * 2 * n_pops));
        // ... but no more than ~3/8 of the CUs' worth of derivation workgroups in all: beyond that they only crowd the hot
        // ones (two populations x 8 walkers: 3 parts = 96 workgroups 21.0 us/step, 4 parts 21.8, 2 parts 23.4; one
        // population x 8 walkers is flat between 5 and 6 parts = 80-96 workgroups)
        const int target = std::max(1, (ctx->n_cu * 3 / 8 + n_walkers * n_pops) / (n_walkers * 2 * n_pops));
        parts = std::max(2, std::min(parts, target));
    }
    sp.derive_parts = std::max(1, std::min(parts, full_parts));
    if (ctx->plan_debug && ctx->plan_debug_key != key * 64 + n_walkers) {
        ctx->plan_debug_key = key * 64 + n_walkers;
        std::fprintf(stderr, "b9 step plan: %d CUs x %d workgroups = %d slots; %d canonical groups of %d tiles; %d walkers x %d hot workgroups (%d groups each) + %d heavy + %d derivation workgroups (%d parts)\n",
                     ctx->n_cu, ctx->step_blocks_per_cu, slots, gr.n_groups, gr.group_tiles, n_walkers, sp.plan.n_blocks * n_pops, sp.plan.groups_per_block,
                     n_walkers * ctx->heavy_parts, n_walkers * 2 * n_pops * sp.derive_parts, sp.derive_parts);
    }
    return sp;
}

// b9_tuning -> the context's plan fields (0 = leave automatic)
void apply_tuning(b9_ctx *ctx, const b9_tuning &t)
{
    ctx->tuning = t;
    ctx->tiles_per_block = std::max(0, t.tiles_per_block);
    ctx->derive_parts = std::max(0, t.derive_parts);
    ctx->derive_order = t.derive_order == 2 ? 0 : (t.derive_order == 3 ? -1 : 1);
    ctx->heavy_parts_fixed = std::max(0, t.heavy_parts);
    ctx->two_launch_steps = t.two_launch_steps != 0;
    ctx->marg_prune = t.marg_no_pruning == 0;
    ctx->timing_group = t.timing_group > 0 ? t.timing_group : 8;
    ctx->plan_debug = t.plan_debug;
    ctx->tree_depth = std::max(0, std::min(B9_TREE_MAX_DEPTH, t.tree_depth));
    if (ctx->marg_piece_units != std::max(0, t.marg_piece_units)) ctx->marg_plan_ok = false;
    ctx->marg_piece_units = std::max(0, t.marg_piece_units);
    ctx->step_occ_key = -1; ctx->plan_debug_key = -1; ctx->tree_occ_key = -1;
}

// The B9_* environment overrides of the same fields (true when any is set).  Parsed once per context, at creation.
bool tuning_from_env(b9_tuning *t)
{
    bool any = false;
    auto num = [&](const char *name, int32_t *dst, bool flip_order = false) {
        const char *v = getenv(name);
        if (!v || !*v) return;
        const int x = atoi(v);
        *dst = flip_order ? (x == 0 ? 2 : (x < 0 ? 3 : 1)) : x;      // B9_DERIVE_ORDER keeps its historical coding (1, 0, < 0)
        any = true;
    };
    num("B9_TILES_PER_BLOCK", &t->tiles_per_block);
    num("B9_DERIVE_PARTS", &t->derive_parts);
    num("B9_DERIVE_ORDER", &t->derive_order, true);
    num("B9_HEAVY_PARTS", &t->heavy_parts);
    num("B9_TWO_LAUNCH_STEPS", &t->two_launch_steps);
    num("B9_MARG_NO_PRUNING", &t->marg_no_pruning);
    num("B9_TIMING_GROUP", &t->timing_group);
    num("B9_PLAN_DEBUG", &t->plan_debug);
    num("B9_TREE_DEPTH", &t->tree_depth);
    num("B9_MARG_PIECE_UNITS", &t->marg_piece_units);
    return any;
}

// Launch plan of the tree-speculative step: the deepest tree (b9_tuning.tree_depth caps or pins it) whose workgroups --
// per walker one writer, 2^d (2^d - 1) candidate derivations in `parts` pieces, (2^d - 1) x heavy_parts heavy-star and
// (2^d - 1) x n_groups hot workgroups -- are ALL resident in one occupancy round, with at most B9_TREE_MAX_GROUPS tile
// groups per node (what one round trip of the walk reads).  depth 1 = none fits: the one-step fused launch runs instead.

TreePlan make_tree_plan(b9_ctx *ctx, int n_walkers, int n_pops)
{
    TreePlan tp{1, 1, 1, 1};
    if (ctx->tree_depth == 1) return tp;
    const Groups gr = make_groups(ctx, n_pops);
    if (gr.n_groups > B9_TREE_MAX_GROUPS) return tp;        // more canonical groups than a walk reads: the one-step launch
    const int key = ((ctx->pk.nfp * 4 + n_pops) * 65536 + ctx->mass_cap) * 2 + (gr.n_groups > 16 * B9_TREE_KD_SMALL ? 1 : 0);
    if (ctx->tree_occ_key != key) {
        int per_cu = 0;
        if (b9k_mcmc_tree_occupancy(ctx->pk, n_pops, ctx->mass_cap, gr.n_groups, &per_cu) != hipSuccess || per_cu < 1) per_cu = 1;
        ctx->tree_blocks_per_cu = per_cu;
        ctx->tree_occ_key = key;
    }
    const long long slots = (long long)ctx->n_cu * ctx->tree_blocks_per_cu;
    const int n_tiles = ctx->st.n_pad / 256;
    const int full_parts = (ctx->mass_cap * (ctx->pk.nfp + 1) + 255) / 256;
    for (int d = B9_TREE_MAX_DEPTH; d >= 2; --d) {
        if (ctx->tree_depth >= 2 && d != ctx->tree_depth) continue;          // pinned
        const long long NN = (1 << d) - 1, NO = 1 << d;
        // one canonical group per hot workgroup; the walk reads a node's partials in one round trip, which bounds their number
        // (the grouping fixes the summation order and is never changed for the tree's sake: no tree then)
        const int tpb = gr.group_tiles, n_groups = gr.n_groups;
        const long long fixed = n_walkers * (1 + NN * ctx->heavy_parts + NN * 8 * ((n_groups * n_pops + 7) / 8));
        const long long per_part = (long long)n_walkers * NO * NN * n_pops;
        const long long room = (long long)(0.95 * slots) - fixed;
        // (the derivation is the launch's longest chain -- decision, parameters, three dependent table round trips -- and more
        //  workgroups per isochrone shorten its last leg: C1 at depth 3, us per chain step: 1 part 7.1, 2: 5.3, 4: 4.7, 6: 4.6)
        int parts = (int)std::min<long long>(std::min(12, full_parts), room / per_part);
        if (ctx->derive_parts > 0) parts = std::min(ctx->derive_parts, full_parts);
        const bool fits = parts >= (d == 2 ? 2 : 1) && fixed + per_part * parts <= slots;
        if (!fits && ctx->tree_depth < 2) continue;          // (a pinned depth runs even when it takes several rounds)
        if (ctx->tree_depth < 2) {
            // Speculation only pays while the chip is under-filled: a depth-d launch evaluates 2^d - 1 nodes for d steps, so once the
            // tiles of a launch saturate the CUs the one-step launch wins. Per-step estimate = (launch floor ~9 us + the larger of
            // ~1.6 us per tile of the longest hot workgroup and ~2.2 us per tile per CU) / d; it reproduces the measured choices:
            // 1 walker x 100k stars d = 3 (10.8 vs 15.4 us/step), x 200k d = 1 (17.4 vs 18.4), x 500k d = 1 (26.2 vs 38.7),
            // 2 walkers x 50k d = 2 (8.4 vs 12.1), 1 x 10k d = 3 (4.4 vs 8.7).
            const B9Groups p1 = make_step_plan(ctx, n_walkers, n_pops).plan;
            const int tpb1 = p1.group_tiles * p1.groups_per_block;
            const double per_cu = 2.2 * (double)n_walkers * n_tiles * n_pops / std::max(1, ctx->n_cu);
            const double est_tree = (9.0 + std::max(1.6 * tpb, per_cu * (double)NN)) / d;
            const double est_step = 9.0 + std::max(1.6 * tpb1, per_cu);
            if (est_tree >= est_step) continue;
        }
        tp.depth = d; tp.group_tiles = tpb; tp.n_groups = n_groups; tp.derive_parts = std::max(1, parts);
        break;
    }
    if (ctx->plan_debug && ctx->plan_debug_key != key * 64 + n_walkers + 1000000 * tp.depth) {
        ctx->plan_debug_key = key * 64 + n_walkers + 1000000 * tp.depth;
        std::fprintf(stderr, "b9 tree plan: %lld slots; depth %d: %d walkers x %d nodes x %d tile groups (%d tiles each), %d derivation parts, %d heavy parts\n",
                     slots, tp.depth, n_walkers, (1 << tp.depth) - 1, tp.n_groups, tp.group_tiles, tp.derive_parts, ctx->heavy_parts);
    }
    return tp;
}

int ensure_tree_buffers(b9_ctx *ctx, int n_walkers, int n_pops, const TreePlan &tp)
{
    const size_t NN = (1u << tp.depth) - 1, NO = 1u << tp.depth;
    const size_t n_cand = (size_t)2 * n_walkers * NO * NN;
    if (n_cand * n_pops > ctx->tree_cand_cap || ctx->tree_iso_stride != ctx->iso_stride) {
        for (void *p : {(void *)ctx->d_tree_hdr, (void *)ctx->d_tree_iso, (void *)ctx->d_tree_par}) if (p) (void)hipFree(p);
        ctx->d_tree_hdr = nullptr; ctx->d_tree_iso = nullptr; ctx->d_tree_par = nullptr; ctx->tree_cand_cap = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_tree_hdr, sizeof(IsoHdr) * n_cand * n_pops));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_tree_iso, sizeof(double) * (size_t)ctx->iso_stride * n_cand * n_pops));
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_tree_par, sizeof(double) * B9_NPARAM * n_cand));
        HIPCHK(ctx, hipMemset(ctx->d_tree_hdr, 0, sizeof(IsoHdr) * n_cand * n_pops));
        ctx->tree_cand_cap = n_cand * n_pops; ctx->tree_iso_stride = ctx->iso_stride;
    }
    const size_t part_stride = ((size_t)tp.n_groups * 4 + ctx->heavy_parts + 1) & ~(size_t)1;
    const size_t n_part = (size_t)2 * n_walkers * NN * part_stride;
    if (n_part > ctx->tree_partial_cap) {
        if (ctx->d_tree_partial) (void)hipFree(ctx->d_tree_partial);
        ctx->d_tree_partial = nullptr; ctx->tree_partial_cap = 0;
        HIPCHK(ctx, hipMalloc((void **)&ctx->d_tree_partial, sizeof(double) * n_part));
        HIPCHK(ctx, hipMemset(ctx->d_tree_partial, 0, sizeof(double) * n_part));
        ctx->tree_partial_cap = n_part;
    }
    return B9_OK;
}

}  // namespace b9i
