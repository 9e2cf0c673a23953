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
// groups_per_block groups: g, g + n_blocks, ...); -1 past the end.
struct TileSeq {
    int g, n_blocks, n_groups, group_tiles, groups_per_block, n_tiles;
    __device__ __forceinline__ int group(int j) const { const int c = g + j * n_blocks; return (j < groups_per_block && c < n_groups) ? c : -1; }
    __device__ __forceinline__ int tile(int c, int t) const { const int k = c + t * n_groups; return (c >= 0 && t < group_tiles && k < n_tiles) ? k : -1; }
};

template <int NFP, int NPOPS>
__device__ __forceinline__ double hot_star(const DevPack &pk, const IsoView<NFP> (&iso)[NPOPS],
                                           double mod, double av, double m1, double q,
                                           const DevStars &st, int il, double log_lam, double log_1ml)
{
    const bool binary = q > 0.0;
    const double m2 = q * m1;
    double ll[2] = {0.0, 0.0};
    // The population loop is deliberately NOT unrolled: unrolled, the compiler overlaps the two
    // populations' row loads and transcendental temporaries and spills (324 B of scratch per lane,
    // 5x slower per star-eval); rolled, the body keeps the single-population register footprint.
    // The isochrone view is picked with wave-uniform selects.
#pragma unroll 1
    for (int k = 0; k < NPOPS; ++k) {
        // Two populations: everything the second pass could share with the first (observations, weights, the
        // per-filter shifts, both brackets' inputs) would be hoisted out of this loop and kept in registers across
        // BOTH passes -- ~60 VGPRs, the difference between two and three waves per SIMD.  The star index and the
        // absorption are passed through an empty asm so that each pass re-derives them (the re-read observations
        // come from the L1 the first pass filled).
        if (NPOPS == 2) { asm volatile("" : "+v"(il)); asm volatile("" : "+v"(av)); }
        const double *is_mass = (NPOPS == 2 && k) ? iso[NPOPS - 1].mass : iso[0].mass;
        const double *is_mags = (NPOPS == 2 && k) ? iso[NPOPS - 1].mags : iso[0].mags;
        const int is_n = (NPOPS == 2 && k) ? iso[NPOPS - 1].n : iso[0].n;
        int lo1, lo2 = 0;
        double t1, t2 = 0.0;
        const double mass0 = is_mass[0];                    // (read once, unconditionally: `||` made it two branches with an LDS round trip each)
        const bool dark1 = !(m1 > 0.0) | (m1 < mass0);
        const bool dark2 = !(m2 > 0.0) | (m2 < mass0);
        find_bracket(is_mass, is_n, m1, lo1, t1);
        if (binary) find_bracket(is_mass, is_n, m2, lo2, t2);
        STAMP(4);
        // two consecutive rows = 2*NFP contiguous doubles
        const double2 *r1 = reinterpret_cast<const double2 *>(is_mags + (size_t)lo1 * NFP);
        const double2 *r2 = reinterpret_cast<const double2 *>(is_mags + (size_t)lo2 * NFP);
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
        const double llk = c0 - 0.5 * (isfinite(chi2) ? chi2 : __builtin_inf());
        if (k == 0) ll[0] = llk; else ll[1] = llk;
    }
    double l = ll[0];
    if (NPOPS == 2) l = logaddexp(log_lam + ll[0], log_1ml + ll[1]);
    return l;       // log p_i L_i ; the field-star mixture is applied by the caller in product form
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
//     its end is requested there, and with one population the lane view lives in scalar registers.
// A star occupies G = 2 NPOPS neighbouring lanes: (population, component) -- star_value_lanes.
template <int NFP, int NPOPS, int NC, class SelectFn>
__device__ __forceinline__ void heavy_stars(const DevPack &pk, const DevStars &st, const IsoHdr *const (&hdr)[NC],
                                            const double *const (&iso_data)[NC], long long iso_stride, int mass_cap,
                                            const double *const (&params)[NC], SelectFn select, bool both_exist, int w, int part, int parts,
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
    // the first chunk's stars (which chunk is a matter of part / wave / lane and of the mode below), also requested now:
    // chunks of PER stars of the descending-mass list are dealt round-robin over the walker's workgroups first and over
    // the waves second
    constexpr int G = 2 * NPOPS, PER = 64 / G;
    const int sub2 = lane % G;
    const int j_a = (part + parts * wave) * PER + lane / G, j_b = (part + parts * (wave & 1)) * PER + lane / G;
    const HeavyStar cur_a = load_heavy_star(st, j_a < st.n ? j_a : st.n - 1);
    const HeavyStar cur_b = NC == 2 ? load_heavy_star(st, j_b < st.n ? j_b : st.n - 1) : cur_a;
    // (both candidates' headers are requested BEFORE the selection is known -- one round trip fewer on the chain; a
    //  candidate that does not exist yet, in the first launch of a block, holds anything: its fields are only used once
    //  selected, and it never is)
    IsoView<NFP> iso[NC][NPOPS];
    bool valid[NC];
    double tip_min[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) valid[c] = load_iso_views<NFP, NPOPS>(hdr[c], iso_data[c], iso_stride, mass_cap, w, iso[c], tip_min[c]);
    // How long a candidate's heavy list is needs no search: the list descends, so "at most T stars above the tip" is
    // heavy_mass[T] <= tip -- one load at an address known at entry.
    // SPECULATIVE mode (the sampler step, both candidates derived): when each candidate's heavy stars fit one round of
    // HALF the workgroups' waves (T = parts * 2 * PER), waves 0-1 evaluate candidate 0 and waves 2-3 candidate 1 side by
    // side, and the role never waits for the decision (the next launch's decision picks the partial of the candidate
    // that counted).  Longer lists wait for the decision and evaluate that candidate alone with all four waves
    // (evaluating both would double the rounds).
    const int T = parts * 2 * PER;
    const double m_first = st.heavy_mass[0], m_T = T < st.n ? st.heavy_mass[T] : -__builtin_inf();
    bool none[NC];                                                               // no star above the candidate's tip
#pragma unroll
    for (int c = 0; c < NC; ++c) none[c] = !(m_first > tip_min[c]);
    bool spec = false;
    (void)m_T; (void)both_exist;
    const int sel = (NC == 2 && !spec) ? select() : 0;
    const int wpc = spec ? 2 : 4;                                                // waves per candidate
    const int cand = spec ? (wave >> 1) : sel;                                   // this wave's candidate
    if (!spec) {
#pragma unroll
        for (int c = 0; c < NC; ++c) if (c != sel) { valid[c] = false; tip_min[c] = __builtin_inf(); }
    }
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
            int v_feh = 0, v_y = 0;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc)
#pragma unroll
                for (int kk = 0; kk < NPOPS; ++kk) {
                    const bool me = (valid[c] ? c : safe) == cc && kp == kk;
                    v_feh = me ? iso[cc][kk].i_feh : v_feh; v_y = me ? iso[cc][kk].i_y : v_y;
                }
            s_axes[tip_lo + e] = pk.tips[(size_t)((v_feh + df) * pk.n_y + (v_y + (dy < ny ? dy : 0))) * na + j];
        }
    }
    __syncthreads();
    HSTAMP(3); B9_MARK("hv-stars-begin");
    // this lane's (candidate, population, component)
    // A star occupies G = 2 NPOPS neighbouring lanes (population, component) of the evaluated candidate's chain.
    const int pop = sub2 >> 1;
    // With one population and the decision taken, everything the lane view holds is the same in every lane of the wave:
    // say so (readfirstlane), and it lives in scalar registers -- ~25 VGPRs in a role that spills for want of them.
    auto uni_i = [](int x) { return (NPOPS == 1) ? __builtin_amdgcn_readfirstlane(x) : x; };
    auto uni_d = [](double x) {
        if (NPOPS != 1) return x;
        const unsigned long long b = __double_as_longlong(x);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        return __longlong_as_double(((unsigned long long)hi << 32) | lo);
    };
    LaneView<NFP> lv;
    {
        const int cs = uni_i((cand ? valid[NC - 1] : valid[0]) ? cand : safe);   // the candidate whose views this lane reads
        const bool B = NPOPS == 2 && pop;
        const bool C = cs != 0;
        // (field-by-field selects with constant indices: a run-time index would put the views in scratch memory)
#define B9_PICK(field) (C ? (B ? iso[NC - 1][NPOPS - 1].field : iso[NC - 1][0].field) : (B ? iso[0][NPOPS - 1].field : iso[0][0].field))
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
    const bool my_valid = uni_i((cand ? valid[NC - 1] : valid[0]) ? 1 : 0) != 0;
    const double my_tip = uni_d(cand ? tip_min[NC - 1] : tip_min[0]);
    double acc = 0.0;
    // chunks of PER stars of the descending-mass list are dealt round-robin over the walker's workgroups first and over a
    // candidate's waves second (a short list spreads over as many CUs as there are parts); a wave stops at its first chunk
    // without a heavy star (masses descend).  Wave-uniform trip count: the shuffles see full EXEC.  The next chunk's star data
    // are requested before this chunk is evaluated.
    int c = part + parts * (spec ? (wave & 1) : wave);
    int j = c * PER + lane / G, jj = j < st.n ? j : st.n - 1;
    HeavyStar cur = spec ? cur_b : cur_a;
    while (c * PER < st.n) {
        const bool live = j < st.n && my_valid && cur.m1 > my_tip;
        if (__ballot(live) == 0ull) break;
        // (the lanes of a star that is NOT above the tip -- most of a short list's last chunk -- present mass 0: "no star",
        //  the shortest path; left alone they would walk the MS/RGB branch, serialised with the live lanes' WD branch)
        HeavyStar hs = cur;
        hs.m1 = live ? cur.m1 : 0.0;
        const double v = star_value_lanes<NFP, NPOPS>(pk, lv, st, jj, hs, sub2);
        if (live && sub2 == 0) {
            if (perstar) perstar[(size_t)w * st.n + st.hv_perm[jj]] = v;
            acc += v;
        }
        c += parts * wpc; j = c * PER + lane / G; jj = j < st.n ? j : st.n - 1;
        cur = load_heavy_star(st, jj);
    }
    HSTAMP(4); B9_MARK("hv-stars-end");
    // the partials: fixed-order sums over the waves (decision first: the evaluated candidate's, 0 in the other slot;
    // speculative: each candidate's two waves)
    {
        const double sum = wave_sum(acc);
        if (lane == 0) s_red[wave] = sum;
    }
    __syncthreads();
    HSTAMP(5);
    if (tid == 0) {
        if (spec) {
            *out_partial[0] = s_red[0] + s_red[1];
            *out_partial[NC - 1] = s_red[2] + s_red[3];
        } else {
            const double tot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
#pragma unroll
            for (int c = 0; c < NC; ++c) *out_partial[c] = c == sel ? tot : 0.0;
        }
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
        heavy_stars<NFP, NPOPS, 1>(pk, st, h1, i1, iso_stride, mass_cap, p1, [] { return 0; }, false, w, part, heavy_parts, o1, perstar, smem);
        return;
    }
    // LDS: the mass column of each population's isochrone -- the bracket search runs in LDS (dependent ds_reads
    // instead of dependent L2 round trips); the magnitude rows a star needs are then read from L2
    // (coalesced: stars are sorted by mass, so neighbouring lanes hit the same or adjacent rows).
    const int tid = threadIdx.x;
    STAMP(0);
    const int L = blockIdx.x - hot_blocks, xcd = L & 7, s = L >> 3;      // hot_blocks is a multiple of 8
    const int w = s % n_walkers;
    const int g = (s / n_walkers) * 8 + xcd;            // this workgroup among the walker's n_blocks
    if (g >= n_blocks) return;
    const TileSeq seq{g, n_blocks, n_groups, group_tiles, groups_per_block, st.n_pad / 256};

    // ---- first round trip: everything that depends only on the kernel arguments ------------
    int i = seq.tile(seq.group(0), 0) * 256 + tid;          // (a group's first tile always exists)
    double m1 = st.mass1[i], q = st.q[i], ea = st.ea[i];
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
    const double log_lam = NPOPS == 2 ? log(lam) : 0.0, log_1ml = NPOPS == 2 ? log1p(-lam) : 0.0;
    STAMP(1);
    __syncthreads();
    STAMP(2);

    for (int j = 0; j < groups_per_block; ++j) {
        const int c = seq.group(j);
        if (c < 0) break;
        MixAcc acc;
        acc.mant = 0.5; acc.expo = 1; acc.add = 0.0;            // = 1.0
        for (int t = 0; t < group_tiles; ++t) {
            const int tile = seq.tile(c, t);
            if (tile < 0) break;
            if (j > 0 || t > 0) {
                i = tile * 256 + tid;
                m1 = st.mass1[i]; q = st.q[i]; ea = st.ea[i];
            }
            STAMP(3);
            if (!valid) {   // outside the grid: the walker's log-posterior is -inf (k_finalize)
                if (perstar && st.perm[i] >= 0) perstar[(size_t)w * st.n + st.perm[i]] = NEG_INF;
                continue;
            }
            if (!(m1 > tip_min)) {     // empty slots hold m1 = +inf
                const double l = hot_star<NFP, NPOPS>(pk, iso, mod, av, m1, q, st, i, log_lam, log_1ml);
                mix_add(acc, ea, l);
                if (perstar) perstar[(size_t)w * st.n + st.perm[i]] = mix_value(ea, l);
            }
        }
        STAMP(7);
        // wave combine (one log per wave); every wave stores its own partial -- no end-of-kernel barrier, so a cheap
        // (single-star) wave never waits for an expensive one
        const double tot = mix_wave_total(acc);
        if ((tid & 63) == 0) partial[(size_t)w * partial_stride + c * 4 + (tid >> 6)] = valid ? tot : 0.0;
    }
    STAMP(8);
}
