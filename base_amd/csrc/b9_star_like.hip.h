// b9_star_like.hip.h -- k_star_like: the given-mass star kernel -- hot (one lane per star) and heavy (WD branch) roles.
// Part of the single translation unit b9_kernels.hip (included there, in this order); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------
// k_star_like  (given-mass mode): the star kernel of b9_logpost; its two roles are also the fused sampler step's
// (k_mcmc_step) and the tree-speculative step's (k_mcmc_tree).
//   HOT workgroups: one lane per star, MS/RGB branch only (hot_star).
//   HEAVY workgroups (they lead the grid): the stars heavier than the walker's AGB tip -- WD branch or NS/BH -- through
//   the general per-star code (heavy_stars), beside the hot ones, so that the WD tables and library transcendentals cost
//   the hot role no registers.
//
// Workgroup -> (star tile, walker) map is XCD-aware: workgroups are dealt round-robin over the 8
// XCDs, so linear id L runs on XCD L % 8.  All walkers of one star tile are given ids with the
// same L % 8 and consecutive L / 8: the tile's star data is fetched from HBM once into that
// XCD's L2 and re-read from L2 by the other walkers.  (Placement affects speed only.)
//
// CANONICAL TILE GROUPS.  The catalogue's tiles (256 slots each) are dealt into n_groups groups of group_tiles tiles:
// group c = tiles c, c + n_groups, c + 2 n_groups, ... (strided over the slot order: binary tiles lead it, so every
// group gets its share of expensive and cheap tiles).  A wave multiplies the field-star mixture factors of its 64 stars
// of the group's tiles in that order and writes ONE partial per group: partial[c * 4 + wave].  n_groups and group_tiles
// are functions of the catalogue, the pack and the device ONLY (make_groups in b9_capi_plan.cpp) -- not of how many walkers
// share the GPU -- so a walker's log-posterior is the same BITS whatever the launch plan: a workgroup takes
// groups_per_block whole groups (g, g + n_blocks, ...), and that number is the plan's only freedom.
// ------------------------------------------------------------------------------------------
// waves per SIMD the star kernels are built for (launch bounds): 3 = up to 168 VGPRs; 16 padded filters need ~250 VGPRs --
// built for three waves that body spills 230-330 B per lane and runs 30 % slower (50k x 16 x 8 walkers: 43.9 vs 30.6 us
// per step) -- so they are built for 2
#define B9_K1_WAVES(NFP, NPOPS) ((NFP) > 8 ? 2 : 3)

// The tile of canonical group c that a workgroup reaches at step t of its tile sequence (group_tiles steps per group,
// groups_per_block groups: g, g + n_blocks, ...); -1 past the end.  `lane_slot` is the lane's slot within a tile: the
// thread id with one population; with two, a workgroup covers HALF of every tile -- slots half * 128 + (tid & 127) -- once
// per population (hot_groups).  A walker has n_blocks * NPOPS hot workgroups: id gb -> block gb / NPOPS, half gb % NPOPS.
struct TileSeq {
    int g, n_blocks, n_groups, group_tiles, groups_per_block, n_tiles, half, lane_slot;
    __device__ __forceinline__ int group(int j) const { const int c = g + j * n_blocks; return (j < groups_per_block && c < n_groups) ? c : -1; }
    __device__ __forceinline__ int tile(int c, int t) const { const int k = c + t * n_groups; return (c >= 0 && t < group_tiles && k < n_tiles) ? k : -1; }
    __device__ __forceinline__ int slot(int tile_) const { return tile_ * 256 + lane_slot; }
};
template <int NPOPS>
__device__ __forceinline__ TileSeq make_tile_seq(int gb, int n_blocks, int n_groups, int group_tiles, int groups_per_block, int n_tiles)
{
    const int tid = threadIdx.x, half = NPOPS == 2 ? (gb & 1) : 0;
    return TileSeq{NPOPS == 2 ? (gb >> 1) : gb, n_blocks, n_groups, group_tiles, groups_per_block, n_tiles, half,
                   NPOPS == 2 ? half * 128 + (tid & 127) : tid};
}

// One star (slot il) on the MS/RGB branch of ONE population's isochrone: log p_i L_i without the mixtures.
template <int NFP>
__device__ __forceinline__ double hot_star(const DevPack &pk, const IsoView<NFP> &iso, double mod, double av, double m1, double q,
                                           const DevStars &st, int il)
{
    const bool binary = q > 0.0;
    const double m2 = q * m1;
    int lo1, lo2 = 0;
    double t1, t2 = 0.0;
    const double mass0 = iso.mass[0];                       // (read once, unconditionally: `||` made it two branches with an LDS round trip each)
    const bool dark1 = !(m1 > 0.0) | (m1 < mass0);
    const bool dark2 = !(m2 > 0.0) | (m2 < mass0);
    find_bracket(iso.mass, iso.n, m1, lo1, t1);
    if (binary) find_bracket(iso.mass, iso.n, m2, lo2, t2);
    STAMP(4);
    // two consecutive rows = 2*NFP contiguous doubles
    const double2 *r1 = reinterpret_cast<const double2 *>(iso.mags + (size_t)lo1 * NFP);
    const double2 *r2 = reinterpret_cast<const double2 *>(iso.mags + (size_t)lo2 * NFP);
    double chi2 = 0.0;
    double2 a1[NFP], a2[NFP];
#pragma unroll
    for (int j = 0; j < NFP; ++j) a1[j] = r1[j];
    if (binary) {
#pragma unroll
        for (int j = 0; j < NFP; ++j) a2[j] = r2[j];
    }
    double p[NFP];
#pragma unroll
    for (int j = 0; j < NFP / 2; ++j) {
        p[2 * j] = dark1 ? B9_MAG_NOFLUX : lerp(a1[j].x, a1[NFP / 2 + j].x, t1);
        p[2 * j + 1] = dark1 ? B9_MAG_NOFLUX : lerp(a1[j].y, a1[NFP / 2 + j].y, t1);
    }
    STAMP(5);
    if (binary) {
#pragma unroll
        for (int j = 0; j < NFP / 2; ++j) {
            const double s0 = dark2 ? B9_MAG_NOFLUX : lerp(a2[j].x, a2[NFP / 2 + j].x, t2);
            const double s1 = dark2 ? B9_MAG_NOFLUX : lerp(a2[j].y, a2[NFP / 2 + j].y, t2);
            p[2 * j] -= (2.5 / LN10) * log1pexp((-0.4 * LN10) * (s0 - p[2 * j]));
            p[2 * j + 1] -= (2.5 / LN10) * log1pexp((-0.4 * LN10) * (s1 - p[2 * j + 1]));
        }
    }
    STAMP(6);
    // observed magnitudes and weights are requested only now: they cost 32 VGPRs while live,
    // and keeping them out of the search / row / combine phases buys a wave per SIMD
    __builtin_amdgcn_sched_barrier(0);
    double obs[NFP], wgt[NFP];
    double c0;
#pragma unroll
    for (int f = 0; f < NFP; ++f) {
        obs[f] = st.obs[B9_SIDX(NFP, f, il)];
        wgt[f] = st.w[B9_SIDX(NFP, f, il)];
    }
    c0 = st.c0[il];
#pragma unroll
    for (int f = 0; f < NFP; ++f) {
        const double d = (p[f] + (mod + pk.abs_m1[f] * av)) - obs[f];
        chi2 = fma(wgt[f] * d, d, chi2);
    }
    return c0 - 0.5 * (isfinite(chi2) ? chi2 : __builtin_inf());
}

// Field-star mixture in PRODUCT form.  sum_i log(A_i + e^{l_i}) = log prod_i (A_i + e^{l_i}),
// A_i = (1 - p_i) fsLike (a per-star constant staged at load): each star costs one exp and one
// multiply; the running product is kept as (mantissa in [0.5,1), binary exponent) so it can neither
// overflow nor underflow, and ONE log per wave turns it back into a sum.  Stars with A_i = 0
// (certain members) or l_i > 600 (e^{l} would overflow; A_i is then negligible) contribute l_i
// additively instead.
struct MixAcc {
    double mant;    // product of factors, renormalised
    int expo;       // its binary exponent
    double add;     // additive part
};

__device__ __forceinline__ void mix_add(MixAcc &a, double ea, double l)
{
    const bool additive = (ea == 0.0) || (l > 600.0);
    const double u = additive ? 1.0 : ea + exp_fast(l);
    a.add += additive ? l : 0.0;
    const double m = a.mant * u;
    a.expo += __builtin_amdgcn_frexp_exp(m);
    a.mant = __builtin_amdgcn_frexp_mant(m);
}

// per-star value for the diagnostic per-star output (library log: u may be < 1)
__device__ __forceinline__ double mix_value(double ea, double l)
{
    return ((ea == 0.0) || (l > 600.0)) ? l : log(ea + exp_fast(l));
}

// wave-wide combine; result valid in lane 0:  log(prod) + sum(add)
__device__ __forceinline__ double mix_wave_total(MixAcc a)
{
    // (lane 0's tree l <- l + 32, + 16, ... + 1 on DPP / permlane moves: b9_common.hip.h lane_down)
#define B9_MIX_STEP(O) {                                                                                   \
        const double m2 = lane_down<O>(a.mant);                                                            \
        const int e2 = lane_down<O>(a.expo);                                                               \
        const double d2 = lane_down<O>(a.add);                                                             \
        const double m = a.mant * m2;                      /* both in [0.5, 1): product in [0.25, 1) */    \
        a.expo += e2 + __builtin_amdgcn_frexp_exp(m);                                                      \
        a.mant = __builtin_amdgcn_frexp_mant(m);                                                           \
        a.add += d2;                                                                                       \
    }
    B9_MIX_STEP(32) B9_MIX_STEP(16) B9_MIX_STEP(8) B9_MIX_STEP(4) B9_MIX_STEP(2) B9_MIX_STEP(1)
#undef B9_MIX_STEP
    // mant in [0.5, 1): log(mant) = log_ge1(2 mant) - ln 2 (the lean log instead of the library one)
    return (log_ge1(a.mant + a.mant) + (double)(a.expo - 1) * 0.693147180559945309417) + a.add;
}

// The hot role's star loop (k_star_like, k_mcmc_step and k_mcmc_tree share it): the workgroup's canonical groups, tile after
// tile (TileSeq), one lane per star and population.  AHEAD (the sampler's kernels): the NEXT tile's star scalars are
// requested before this tile's arithmetic -- across group boundaries too: one memory round trip less on every tile after
// the first, for 6 VGPRs (k_star_like, which also carries the per-star output, has none to spare).  At a
// group's end every wave combines its 64 products (one log per wave) and lane 0 hands the partial of its chunk k (the
// 64-slot quarter of a tile) to store(c, k, total) -- no end-of-kernel barrier with one population, so a cheap
// (single-star) wave never waits for an expensive one.  (i, m1, q, ea) are the first tile's scalars, requested by the caller
// at its entry.  `valid` is uniform over the workgroup (the walker's candidate is inside the grid).
//
// TWO POPULATIONS are laid over the workgroup's WAVES, not looped over in a lane: the workgroup covers half of every tile
// (TileSeq), waves 0-1 evaluate its 128 stars on population A's isochrone, waves 2-3 the same stars on population B's.  A
// wave's isochrone is wave-uniform and the body keeps the one-population register footprint and instruction stream; B's
// value crosses through LDS (two buffers, one barrier per tile) and the A lane, which owns the star, forms the one
// logaddexp, the field-star factor and the partial.  Which lane multiplies which star into which partial is the
// one-population mapping: partial (c, k) is the product over chunk k of the group's tiles, whoever computes it.
__device__ __forceinline__ double *hot_lds_cross() { __shared__ double s_ll[2][128]; return &s_ll[0][0]; }

template <int NFP, int NPOPS, bool AHEAD, class Store>
__device__ __forceinline__ void hot_groups(const DevPack &pk, const DevStars &st, const TileSeq &seq, const IsoView<NFP> (&iso)[NPOPS],
                                           bool valid, double tip_min, double mod, double av, double log_lam, double log_1ml,
                                           int i, double m1, double q, double ea, double *__restrict__ perstar_row, Store store)
{
    const int tid = threadIdx.x;
    const bool pop_b = NPOPS == 2 && __builtin_amdgcn_readfirstlane(tid >> 7) != 0;      // (owner lanes: population A's)
    const int k = NPOPS == 2 ? seq.half * 2 + ((tid >> 6) & 1) : (tid >> 6);            // this wave's chunk of a tile
    if (!valid) {                                           // outside the grid: the walker's log-posterior is -inf
        for (int j = 0; j < seq.groups_per_block; ++j) {
            const int c = seq.group(j);
            if (c < 0) break;
            if (perstar_row && !pop_b)
                for (int t = 0; t < seq.group_tiles && seq.tile(c, t) >= 0; ++t) {
                    const int s = st.perm[seq.slot(seq.tile(c, t))];
                    if (s >= 0) perstar_row[s] = NEG_INF;
                }
            if ((tid & 63) == 0 && !pop_b) store(c, k, 0.0);
        }
        return;
    }
    IsoView<NFP> mine = iso[0];
    if (pop_b) mine = iso[NPOPS - 1];
    int j = 0, t = 0, c = seq.group(0), n = 0;
    MixAcc acc;
    acc.mant = 0.5; acc.expo = 1; acc.add = 0.0;            // = 1.0
    while (c >= 0) {
        int jn = j, tn = t + 1, cn = c;                     // the workgroup's next tile, across the group boundary
        if (seq.tile(c, tn) < 0) { jn = j + 1; tn = 0; cn = seq.group(jn); }
        const int i_n = seq.slot(cn >= 0 ? seq.tile(cn, tn) : seq.tile(c, t));
        double m1_n, q_n, ea_n;
        if (AHEAD) { m1_n = st.mass1[i_n]; q_n = st.q[i_n]; ea_n = st.ea[i_n]; }
        STAMP(3);
        const bool live = !(m1 > tip_min);                  // (else: heavier than the AGB tip -- the heavy role's -- or an empty slot, +inf)
        double l = 0.0;
        if (live) l = hot_star<NFP>(pk, mine, mod, av, m1, q, st, i);
        if constexpr (NPOPS == 2) {
            double *const cross = hot_lds_cross() + (n & 1) * 128;
            if (pop_b) cross[tid & 127] = l;
            __syncthreads();
            if (!pop_b) l = logaddexp(log_lam + l, log_1ml + cross[tid & 127]);
        }
        if (live && !pop_b) {
            mix_add(acc, ea, l);
            if (perstar_row) perstar_row[st.perm[i]] = mix_value(ea, l);
        }
        if (cn != c) {                                      // the group is complete
            STAMP(7);
            if (!pop_b) {
                const double tot = mix_wave_total(acc);
                if ((tid & 63) == 0) store(c, k, tot);
            }
            acc.mant = 0.5; acc.expo = 1; acc.add = 0.0;
        }
        if (!AHEAD) { m1_n = st.mass1[i_n]; q_n = st.q[i_n]; ea_n = st.ea[i_n]; }
        i = i_n; m1 = m1_n; q = q_n; ea = ea_n; j = jn; t = tn; c = cn; ++n;
    }
}

// The stars the hot path skips -- primary heavier than the walker's AGB tip (IFMR -> WD cooling -> WD
// atmosphere, or NS/BH) -- are evaluated by extra workgroups of the SAME launch (`parts` per walker), through the
// general per-star code.  The role is ONE long dependent chain per star (~10 us: precursor age -> cooling age ->
// cooling tracks -> atmosphere rows), so it is built to start that chain at once:
//   * In the fused sampler step (NC = 2 candidate rows) everything it stages is requested for BOTH candidates at entry,
//     in the same memory round trip as the previous step's accept/reject; only the candidate that decision selects is then
//     evaluated (evaluating both cost more than it saved: usually only one of them has heavy stars at all).
//   * It does NOT search for the number of heavy stars: stars are also listed by descending mass (heavy_mass and the
//     hv_* copies of their data), a wave walks that list in chunks and stops at the first chunk in which no star of the
//     evaluated candidate is above its AGB tip ("none at all" is one load: heavy_mass[0] <= tip).
//   * Everything it stages in LDS -- the searched axes (packed into one run at load time, DevPack::heavy_const), the
//     AGB-tip table, per (candidate, population) the mass column, per candidate the parameter row -- is requested at
//     entry, segment by segment, all loads independent: one memory round trip (it used to be 6 + 4 NPOPS dependent ones).
//   * It is short of REGISTERS (it wants ~190 VGPRs under the kernel's cap of 168): what the chain does not need until
//     its end is requested there, and the lane view -- one candidate, one population per wave -- lives in scalar registers.
// A star occupies a PAIR of neighbouring lanes (its two components: star_ll_lanes).  TWO POPULATIONS are two wave pairs:
// waves 0-1 walk the chunks on population A's isochrone, waves 2-3 the SAME chunks on population B's, so a wave's lane view
// is uniform (scalar registers) and a star between the two populations' AGB tips has its WD chain in one wave and its MS/RGB
// bracket in another instead of both, serialised, in every wave.  B's log-likelihood crosses through LDS; the A lane forms
// the mixtures.  The chunk loop's trip count is then agreed by the whole workgroup (__syncthreads_or).
template <int NFP, int NPOPS, int NC, class SelectFn>
__device__ __forceinline__ void heavy_stars(const DevPack &pk, const DevStars &st, const IsoHdr *const (&hdr)[NC],
                                            const double *const (&iso_data)[NC], long long iso_stride, int mass_cap,
                                            const double *const (&params)[NC], SelectFn select, int w, int part, int parts,
                                            double *const (&out_partial)[NC], double *__restrict__ perstar, double *smem)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *s_red = smem;                                // [NC][4]
    double *s_axes = smem + 8;
    // ---- the staged list, pass 1: everything whose address needs no isochrone header -- requested at entry, in the same
    // memory round trip as the headers.  Segments: NC0 common axes; the whole AGB-tip table when it is small (else the
    // corner columns follow in pass 2); each candidate's parameter row; per (candidate, population) the derived isochrone's mass column (a heavy
    // primary's companion, or a star that is heavy under one candidate only, is on the MS/RGB branch: its bracket search
    // then runs in LDS, as the hot role's does).
    // segments: 0 = the packed axes (DevPack::heavy_const), 1 = AGB tips, then the mass columns, then the parameter rows
    constexpr int NC0 = 1, NM0 = NC0 + 1, NP0 = NM0 + NPOPS * NC, NSEG = NP0 + NC;
    const int na = pk.n_age, ny = pk.n_y > 1 ? 2 : 1, n_tips = pk.n_feh * pk.n_y * na;
    const bool stage_wc_age = pk.hc_age_staged != 0;                                                 // (as heavy_lds_doubles sized the LDS)
    const bool tips_all = n_tips <= B9_TIPS_LDS_MAX;
    const double *seg_src[NSEG];
    int seg_off[NSEG + 1];
    {
        seg_off[0] = 0;
        seg_src[0] = pk.heavy_const; seg_off[1] = pk.hc_len;
        seg_src[NC0] = pk.tips; seg_off[NM0] = seg_off[NC0] + (tips_all ? n_tips : 4 * NPOPS * NC * na);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int kp = 0; kp < NPOPS; ++kp) {         // (a candidate's buffer exists even when its contents do not yet)
                const int k = NM0 + c * NPOPS + kp;
                seg_src[k] = iso_data[c] + (size_t)(w * NPOPS + kp) * iso_stride;
                seg_off[k + 1] = seg_off[k] + mass_cap;
            }
        // ... and each candidate's parameter row: the chain reads a parameter here and there, behind branches the
        // compiler cannot lift a global load over -- each was its own memory round trip on the chain
#pragma unroll
        for (int c = 0; c < NC; ++c) { seg_src[NP0 + c] = params[c] + (size_t)w * B9_NPARAM; seg_off[NP0 + c + 1] = seg_off[NP0 + c] + B9_NPARAM; }
    }
    const int tip_lo = seg_off[NC0];
    // Segment by segment, the first RC[g] x 256 elements of each travel through registers (requested here, written to LDS
    // after the decision): a thread's element of chunk i of segment g is seg_src[g][tid + 256 i] -- no search for "which
    // segment is element e in" (the flat-list version spent ~40 VALU instructions per element on that: 16 elements per
    // thread with two populations, 2.3 us of the role's entry).  Longer segments finish in a loop after the decision.
    // Pass 1 skips the AGB-tip columns of large grids: their addresses need the headers (pass 2).
    auto rc = [](int g) constexpr { return g == 0 ? 2 : (g == NC0 ? 8 : (g >= NM0 && g < NP0 ? 2 : 1)); };      // chunks in registers
    constexpr int NSV = 2 + 8 + 2 * NPOPS * NC + NC;
    double sv[NSV];
    {
        int idx = 0;
#pragma unroll
        for (int g = 0; g < NSEG; ++g) {
            const int len = (g == NC0 && !tips_all) ? 0 : seg_off[g + 1] - seg_off[g];
#pragma unroll
            for (int i = 0; i < rc(g); ++i, ++idx) {
                const int e = tid + 256 * i;
                sv[idx] = 0.0;
                if (len > 256 * i) sv[idx] = seg_src[g][e < len ? e : len - 1];          // (clamped, not branched per lane)
            }
        }
    }
    auto stage_store = [&](const double (&v)[NSV]) {
        int idx = 0;
#pragma unroll
        for (int g = 0; g < NSEG; ++g) {
            const int len = (g == NC0 && !tips_all) ? 0 : seg_off[g + 1] - seg_off[g];
#pragma unroll
            for (int i = 0; i < rc(g); ++i, ++idx) {
                const int e = tid + 256 * i;
                if (e < len) s_axes[seg_off[g] + e] = v[idx];
            }
            for (int e = tid + 256 * rc(g); e < len; e += 256) s_axes[seg_off[g] + e] = seg_src[g][e];     // very long segments only
        }
    };
    // the first chunk's stars, also requested now: chunks of PER stars of the descending-mass list are dealt round-robin
    // over the walker's workgroups first and over a population's waves second
    constexpr int PER = 32, WPP = NPOPS == 2 ? 2 : 4;           // stars per chunk (a lane pair each); waves per population
    const int comp = lane & 1;
    const int wv = NPOPS == 2 ? (wave & 1) : wave;               // this wave among its population's
    const bool pop_b = NPOPS == 2 && __builtin_amdgcn_readfirstlane(wave >> 1) != 0;
    const int j_a = (part + parts * wv) * PER + lane / 2;
    const int jj_a = j_a < st.n ? j_a : st.n - 1;
    const double cur_m1_a = st.heavy_mass[jj_a], cur_q_a = st.hv_q[jj_a];
    const int cur_flags_a = st.hv_flags[jj_a];
    // (both candidates' headers are requested BEFORE the selection is known -- one round trip fewer on the chain; a
    //  candidate that does not exist yet, in the first launch of a block, holds anything: its fields are only used once
    //  selected, and it never is)
    // A wave keeps ONE population's view per candidate (its own), and of the other population only what decides
    // validity and the smaller tip: four full views would not fit the scalar registers.
    IsoView<NFP> mine[NC];
    bool valid[NC];
    double tip_min[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        valid[c] = true; tip_min[c] = __builtin_inf();
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) {
            const IsoHdr h = hdr[c][w * NPOPS + kp];
            valid[c] = valid[c] && h.valid;
            tip_min[c] = h.agb_tip < tip_min[c] ? h.agb_tip : tip_min[c];
            if (kp == 0 || pop_b) {                          // (wave-uniform: population B's waves overwrite A's view with their own)
                const double *g = iso_data[c] + (size_t)(w * NPOPS + kp) * iso_stride;
                mine[c].n = h.n; mine[c].tip = h.agb_tip; mine[c].i_feh = h.i_feh; mine[c].i_y = h.i_y;
                mine[c].t_feh = h.t_feh; mine[c].t_y = h.t_y; mine[c].mass = g; mine[c].mags = g + mass_cap;
            }
        }
    }
    // Whether a candidate has a heavy star at all needs no search: the list descends, so "none" is heavy_mass[0] <= tip --
    // one load at an address known at entry.  The role waits for the decision and evaluates that candidate alone
    // (evaluating both cost more than it saved: usually only one of them has heavy stars at all).
    const double m_first = st.heavy_mass[0];
    bool none[NC];                                                               // no star above the candidate's tip
#pragma unroll
    for (int c = 0; c < NC; ++c) none[c] = !(m_first > tip_min[c]);
    const int sel = NC == 2 ? select() : 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) if (c != sel) { valid[c] = false; tip_min[c] = __builtin_inf(); }
    {   // nothing to do: no valid candidate, or no heavy star under any candidate this workgroup evaluates
        bool any = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) any = any || (valid[c] && !none[c]);
        if (!any) {
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < NC; ++c) *out_partial[c] = 0.0;
            }
            return;
        }
    }
    const int safe = valid[0] ? 0 : NC - 1;                                      // a candidate whose views are real
    HSTAMP(1);
    stage_store(sv);
    if (!tips_all) {
        // pass 2 (large grids only): the AGB-tip columns of each (candidate, population)'s four (FeH, Y) corners, whose
        // addresses the headers give
        for (int e = tid; e < 4 * NPOPS * NC * na; e += 256) {
            const int col = e / na, j = e - col * na, q = col & 3, ck = col >> 2, c = ck / NPOPS, kp = ck - c * NPOPS;
            const int df = q >> 1, dy = q & 1;
            const IsoHdr *hp = (valid[c] ? c : safe) ? hdr[NC - 1] : hdr[0];     // (re-read: this path runs on large grids only)
            const int v_feh = hp[w * NPOPS + kp].i_feh, v_y = hp[w * NPOPS + kp].i_y;
            s_axes[tip_lo + e] = pk.tips[(size_t)((v_feh + df) * pk.n_y + (v_y + (dy < ny ? dy : 0))) * na + j];
        }
    }
    __syncthreads();
    HSTAMP(3); B9_MARK("hv-stars-begin");
    // Everything the lane view holds is the same in every lane of the wave (one candidate, one population per wave): say
    // so (readfirstlane), and it lives in scalar registers -- ~25 VGPRs in a role that is short of them.
    auto uni_i = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    auto uni_d = [](double x) { return wave_uniform(x); };
    LaneView<NFP> lv;
    {
        const int cs = uni_i((sel ? valid[NC - 1] : valid[0]) ? sel : safe);     // the candidate whose views this wave reads
        const bool B = pop_b;
        const bool C = cs != 0;
        // (field-by-field selects with constant indices: a run-time index would put the views in scratch memory)
#define B9_PICK(field) (C ? mine[NC - 1].field : mine[0].field)
        lv.is_mags = B9_PICK(mags); lv.is_n = uni_i(B9_PICK(n)); lv.is_tip = uni_d(B9_PICK(tip));
        lv.t_feh = uni_d(B9_PICK(t_feh)); lv.t_y = uni_d(B9_PICK(t_y));
        const int v_feh = uni_i(B9_PICK(i_feh)), v_y = uni_i(B9_PICK(i_y));
#undef B9_PICK
        const int ck = (C ? NC - 1 : 0) * NPOPS + (B ? NPOPS - 1 : 0);
        lv.is_mass = s_axes + seg_off[NM0] + ck * mass_cap;                      // the LDS copy of the mass column
        lv.par = s_axes + seg_off[NP0] + (C ? NC - 1 : 0) * B9_NPARAM;           // the LDS copy of the candidate's parameter row
        lv.ax.log_age = s_axes + pk.hc_off[0];
        lv.ax.wc_log_age_lds = stage_wc_age ? s_axes + pk.hc_off[1] : nullptr;
        lv.ax.wc_mass = s_axes + pk.hc_off[2]; lv.ax.wc_carb = s_axes + pk.hc_off[3];
        lv.ax.at_log_teff = s_axes + pk.hc_off[4]; lv.ax.at_logg = s_axes + pk.hc_off[5]; lv.ax.wc_track = s_axes + pk.hc_off[6];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int df = q >> 1, dy = q & 1;
            const int col = tips_all ? ((v_feh + df) * pk.n_y + (v_y + (dy < ny ? dy : 0))) : ck * 4 + q;
            lv.ax.tips[q] = s_axes + tip_lo + col * na;
        }
    }
    const bool my_valid = uni_i((sel ? valid[NC - 1] : valid[0]) ? 1 : 0) != 0;
    const double my_tip = uni_d(sel ? tip_min[NC - 1] : tip_min[0]);             // (the smaller of the populations' tips: the same in every wave)
    double acc = 0.0;
    // A wave stops at its first chunk without a heavy star (masses descend) -- with two populations, when no wave of the
    // workgroup has one (the pairs exchange through LDS behind the barrier that also takes that vote).  Wave-uniform trip
    // count: the shuffles see full EXEC.  The next chunk's star data are requested before this chunk is evaluated.
    int c = part + parts * wv;
    int j = c * PER + lane / 2, jj = j < st.n ? j : st.n - 1;
    double cur_m1 = cur_m1_a, cur_q = cur_q_a;                                   // (scalars, not a HeavyStar: a copied struct left a dead stack slot)
    int cur_flags = cur_flags_a;
    for (int it = 0; ; ++it) {
        const bool live = c * PER < st.n && j < st.n && my_valid && cur_m1 > my_tip;
        const bool any = __ballot(live) != 0ull;
        if (NPOPS == 1 && !any) break;
        // (the lanes of a star that is NOT above the tip -- most of a short list's last chunk -- present mass 0: "no star",
        //  the shortest path; left alone they would walk the MS/RGB branch, serialised with the live lanes' WD branch)
        HeavyStar hs;
        hs.m1 = live ? cur_m1 : 0.0; hs.q = cur_q; hs.flags = cur_flags;
        double l = 0.0;
        if (any) l = star_ll_lanes<NFP>(pk, lv, st, jj, hs, comp);
        if constexpr (NPOPS == 2) {
            __shared__ double s_cross[2][128];
            double *const cross = s_cross[it & 1];
            if (pop_b) cross[wv * 64 + lane] = l;
            if (!__syncthreads_or(any ? 1 : 0)) break;                          // (no wave of the workgroup has a heavy star left)
            if (!pop_b) {                                                        // (log lambda is formed here, after the chain: nothing rides through it)
                const double lam = lv.par[B9_P_LAMBDA];
                l = logaddexp(log(lam) + l, log1p(-lam) + cross[wv * 64 + lane]);
            }
        }
        if (live && comp == 0 && !pop_b) {
            const double v = logaddexp(st.hv_la[jj], l);
            if (perstar) perstar[(size_t)w * st.n + st.hv_perm[jj]] = v;
            acc += v;
        }
        c += parts * WPP; j = c * PER + lane / 2; jj = j < st.n ? j : st.n - 1;
        cur_m1 = st.heavy_mass[jj]; cur_q = st.hv_q[jj]; cur_flags = st.hv_flags[jj];
    }
    HSTAMP(4); B9_MARK("hv-stars-end");
    // the partial: a fixed-order sum over the waves -- the evaluated candidate's, 0 in the other slot
    {
        const double sum = wave_sum(acc);
        if (lane == 0) s_red[wave] = sum;
    }
    __syncthreads();
    HSTAMP(5);
    if (tid == 0) {
        const double tot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
#pragma unroll
        for (int c2 = 0; c2 < NC; ++c2) *out_partial[c2] = c2 == sel ? tot : 0.0;
    }
}

template <int NFP, int NPOPS>
__global__ __launch_bounds__(256, B9_K1_WAVES(NFP, NPOPS)) void k_star_like(DevPack pk, DevStars st,
                                                    const IsoHdr *__restrict__ hdr,
                                                    const double *__restrict__ iso_data,
                                                    long long iso_stride, int mass_cap,
                                                    const double *__restrict__ params, int n_walkers,
                                                    double *__restrict__ partial, long long partial_stride,
                                                    int n_groups, int group_tiles, int groups_per_block, int n_blocks,
                                                    double *__restrict__ perstar, int hot_blocks, int heavy_parts)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // Heavy-star workgroups come FIRST in the grid (hot_blocks = their padded count): they have the
    // longest dependent chain, so they must start at once and run beside the hot workgroups.
    if ((int)blockIdx.x < hot_blocks) {        // hot_blocks doubles as "first hot workgroup id"
        const int hb = blockIdx.x;
        if (hb >= n_walkers * heavy_parts) return;            // padding to a multiple of 8
        const int w = hb / heavy_parts, part = hb - w * heavy_parts;
        const IsoHdr *const h1[1] = {hdr};
        const double *const i1[1] = {iso_data}, *const p1[1] = {params};
        double *const o1[1] = {partial + (size_t)w * partial_stride + (size_t)n_groups * 4 + part};
        heavy_stars<NFP, NPOPS, 1>(pk, st, h1, i1, iso_stride, mass_cap, p1, [] { return 0; }, w, part, heavy_parts, o1, perstar, smem);
        return;
    }
    // LDS: the mass column of each population's isochrone -- the bracket search runs in LDS (dependent ds_reads
    // instead of dependent L2 round trips); the magnitude rows a star needs are then read from L2
    // (coalesced: stars are sorted by mass, so neighbouring lanes hit the same or adjacent rows).
    const int tid = threadIdx.x;
    STAMP(0);
    const int L = blockIdx.x - hot_blocks, xcd = L & 7, s = L >> 3;      // hot_blocks is a multiple of 8
    const int w = s % n_walkers;
    const int g = (s / n_walkers) * 8 + xcd;            // this workgroup among the walker's n_blocks * NPOPS
    if (g >= n_blocks * NPOPS) return;
    const TileSeq seq = make_tile_seq<NPOPS>(g, n_blocks, n_groups, group_tiles, groups_per_block, st.n_pad / 256);

    // ---- first round trip: everything that depends only on the kernel arguments ------------
    const int i = seq.slot(seq.tile(seq.group(0), 0));      // (a group's first tile always exists)
    const double m1 = st.mass1[i], q = st.q[i], ea = st.ea[i];
    // the mass columns: the source address needs no header field, and copying the full capacity
    // instead of hdr.n entries costs nothing (the tail is never searched)
    double *const lds_mass = smem;
    for (int c = 0; c < NPOPS; ++c) {
        const double2 *sm = reinterpret_cast<const double2 *>(iso_data + (size_t)(w * NPOPS + c) * iso_stride);
        double2 *dm = reinterpret_cast<double2 *>(lds_mass + (size_t)c * mass_cap);
        for (int j = tid; j < mass_cap / 2; j += 256) dm[j] = sm[j];
    }
    // headers and the few parameters the star loop needs (scalar loads, same round trip)
    IsoView<NFP> iso[NPOPS];
    const double *par = params + (size_t)w * B9_NPARAM;
    bool valid = true;
    double tip_min = __builtin_inf();
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const IsoHdr h = hdr[w * NPOPS + kp];
        valid = valid && h.valid;
        iso[kp].n = h.n; iso[kp].tip = h.agb_tip;
        iso[kp].i_feh = h.i_feh; iso[kp].i_y = h.i_y; iso[kp].t_feh = h.t_feh; iso[kp].t_y = h.t_y;
        iso[kp].mass = lds_mass + (size_t)kp * mass_cap;
        iso[kp].mags = iso_data + (size_t)(w * NPOPS + kp) * iso_stride + mass_cap;
        tip_min = h.agb_tip < tip_min ? h.agb_tip : tip_min;
    }
    const double mod = par[B9_P_MOD], av = par[B9_P_ABS], lam = NPOPS == 2 ? par[B9_P_LAMBDA] : 1.0;
    const double log_lam = NPOPS == 2 ? wave_uniform(log(lam)) : 0.0, log_1ml = NPOPS == 2 ? wave_uniform(log1p(-lam)) : 0.0;
    STAMP(1);
    __syncthreads();
    STAMP(2);
    double *const prow = partial + (size_t)w * partial_stride;
    hot_groups<NFP, NPOPS, false>(pk, st, seq, iso, valid, tip_min, mod, av, log_lam, log_1ml, i, m1, q, ea,
                           perstar ? perstar + (size_t)w * st.n : nullptr,
                           [&](int c, int k, double tot) { prow[c * 4 + k] = tot; });
    STAMP(8);
}
