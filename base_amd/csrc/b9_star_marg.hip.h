// b9_star_marg.hip.h -- k_star_marg: marginalised mode, one wavefront per star (+ the sampleMass draws).
// Part of the single translation unit b9_kernels.hip (included there, in this order); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------
// k_star_marg  (marginalised mode; SURVEY 8a row a6 "marg.cpp-like", [RECALL] margEvolveWithBinary)
//
// ONE WAVEFRONT PER STAR.  The star's likelihood is integrated over primary mass (iso_increm equal
// sub-steps inside every EEP interval of the derived isochrone, left-endpoint rule) and mass ratio
// (n_q nodes j / n_q); every node contributes exp(ll) dM / n_q to a per-lane online log-sum-exp, the 64
// lanes are combined with wavefront shuffles, and the star's value is written to its slot (the per-walker
// sum over stars is k_finalize's fixed-order block sum).
//
// Round 3 layout.  WORKGROUPS ARE PERSISTENT over a walker's stars: the walker's isochrone (mass column +
// magnitude rows) and the chunk-bound table are staged in LDS ONCE per workgroup; its four waves then walk
// the star slots in strides of the launch's wave count (each wave on its own: no workgroup barrier after
// the staging).  Lane = primary-mass node of a 64-node chunk.  NOTHING about a node depends on the star: its primary
// magnitudes, its log(prior dM / n_q), and -- per mass ratio j / n_q -- the COMBINED magnitudes of the node with that
// companion are tabulated once per call (k_marg_table; L2-resident: the workgroups of a walker share an XCD).  A star's
// work is then nothing but chi^2 sums: per node the single star's (j = 0) and one per companion.
// (Round 2's kernel re-derived all of that per star: bracket searches, interpolations, an exponential and a logarithm per
// flux combine -- 9090 VALU wave-instructions per star-eval; its LDS rows at a 64-byte stride were 4-way bank-conflicted,
// a padded stride removed the conflicts without moving the time, and the rows then left LDS altogether.)
// A star of stage WD integrates over (AGB tip, M_wd_up] in 8 iso_increm steps through the WD branch: those stars (a few
// per cent of a cluster, listed at load time: DevStars::wd_slot) have a kernel of their own, k_star_marg_wd, so that
// the WD branch's registers (it alone wants > 200 VGPRs) do not set the occupancy of every other star's evaluation.
// ------------------------------------------------------------------------------------------
struct Lse { double mx, sm; };      // online log-sum-exp:  value = mx + log(sm)
#ifndef B9_MARG_CUT
#define B9_MARG_CUT 40.0             // nodes more than this many e-folds below the running maximum are dropped
#endif
__device__ __forceinline__ void lse_add(Lse &a, double x)
{
    if (x == NEG_INF) return;
    if (x > a.mx) { a.sm = a.sm * exp_fast(a.mx - x) + 1.0; a.mx = x; }
    else a.sm += exp_fast(x - a.mx);
}

__device__ __forceinline__ Lse lse_merge(Lse a, Lse b)
{
    if (b.mx == NEG_INF) return a;
    if (a.mx == NEG_INF) return b;
    Lse r;
    if (a.mx >= b.mx) { r.mx = a.mx; r.sm = a.sm + b.sm * exp_fast(b.mx - a.mx); }
    else { r.mx = b.mx; r.sm = b.sm + a.sm * exp_fast(a.mx - b.mx); }
    return r;
}

// log(x) for any positive normal x: log_ge1's reduction and polynomial are fdlibm's general e_log.c form (k may be
// negative), 1 ulp; the library log / log10 (98 VALU instructions each) were a fifth of the primary pass.
__device__ __forceinline__ double log_pos(double x) { return log_ge1(x); }

__device__ __forceinline__ double log_prior_mass_dev(double lmn, double m)
{
    const double lm = log_pos(m);
    const double z = (lm * (1.0 / LN10) - MF_MU) / MF_SIGMA;
    return lmn - 0.5 * z * z - lm - log(LN10);
}

// SAMPLE (b9_sample_mass, the sampleMass counterpart -- SURVEY 8f row 4): besides the marginal, every
// star draws ONE (primary mass, mass ratio[, population]) node from its conditional posterior over the
// same grid by the Gumbel-max rule: the node that maximises  log-term + G,  G = -log(-log u),
// u = Philox(seed; row, star, node).  The rule is an argmax, hence independent of the order in
// which lanes visit the nodes -- the CPU oracle, which walks them sequentially, picks the same node.
// Nodes the pruning drops (> 40 e-folds below the maximum) draw no number: they could only win with
// probability e^-40.
#ifdef B9_MARG_STATS      // diagnostic build only: where the marginalised kernel's iterations go
__device__ unsigned long long g_marg_stats[8];
#define MSTAT(k, v) do { if (lane == 0) atomicAdd(&g_marg_stats[k], (unsigned long long)(v)); } while (0)
extern "C" int b9_debug_marg_stats(unsigned long long *out, int clear)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_marg_stats), sizeof(unsigned long long) * 8);
    if (clear) { unsigned long long z[8] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_marg_stats), z, sizeof z); }
    return rc;
}
#else
#define MSTAT(k, v) do {} while (0)
#endif

struct MargSample {
    double *mass, *ratio, *member;   // [rows][n_stars]
    int *pop;                        // [rows][n_stars] or null
    unsigned k0, k1;
    long long row0;                  // global index of row 0 (RNG counter)
};

struct Best { double key, mass, ratio; int pop; };

__device__ __forceinline__ double gumbel(unsigned k0, unsigned k1, unsigned long long row, unsigned star, unsigned long long node, unsigned pop)
{
    unsigned r[4];
    philox4x32((unsigned)row, star, (unsigned)node, (unsigned)(node >> 32) * 2u + pop, k0, k1 ^ (unsigned)(row >> 32), r);
    return -log(-log(u01(r[0], r[1])));
}

// ordering point for this wave's own LDS traffic (a wave's LDS operations execute in order; the fences keep the
// compiler from moving the reads above the writes)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Waves per SIMD the instances are built for (tools/kernel_resources.py; every instance at <= 16 B of scratch per lane).
#ifndef B9_MARG_WAVES
#define B9_MARG_WAVES(NFP, NPOPS, SAMPLE) (((NFP) >= 16 || (SAMPLE)) ? 2 : 3)
#endif

// doubles of per-wave LDS scratch: the star's shift / obs / weight per filter
#define B9_MARG_WAVE_SCRATCH(NFP) (3 * (NFP))

template <int NFP, int NPOPS, bool SAMPLE>
__global__ __launch_bounds__(256, B9_MARG_WAVES(NFP, NPOPS, SAMPLE)) void k_star_marg(DevPack pk, DevStars st, const IsoHdr *__restrict__ hdr,
                                                    const double *__restrict__ iso_data, long long iso_stride,
                                                    int mass_cap, const double *__restrict__ params,
                                                    double *__restrict__ vals, double *__restrict__ perstar,
                                                    int K, int Q, MargSample ms, int chunk_cap,
                                                    const double *__restrict__ tab, long long tab_stride, int npad,
                                                    int n_walkers, int wg_per_walker)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid of n_walkers * wg_per_walker workgroups.  Workgroup ids are dealt round-robin over the 8 XCDs: with a
    // multiple of 8 walkers every walker's workgroups are given ids of ONE residue mod 8, so its companion table and
    // isochrone are fetched into one XCD's L2 (speed only; any placement is correct).
    int w, bx;
    if ((n_walkers & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3, wpx = n_walkers >> 3;
        w = xcd + 8 * (i % wpx); bx = i / wpx;
    } else { w = blockIdx.x / wg_per_walker; bx = blockIdx.x - w * wg_per_walker; }
    const double *par = params + (size_t)w * B9_NPARAM;
    IsoView<NFP> iso_g[NPOPS];                           // the derived isochrones in global memory (rows NFP apart)
    double tip_min;
    const bool valid = load_iso_views<NFP, NPOPS>(hdr, iso_data, iso_stride, mass_cap, w, iso_g, tip_min);
    const int wave0 = bx * 4 + wave, wave_stride = wg_per_walker * 4;
    if (!valid) {
        for (int slot = wave0; slot < st.n_pad; slot += wave_stride)
            if (lane == 0) {
                vals[(size_t)w * st.n_pad + slot] = 0.0;
                if (perstar && st.perm[slot] >= 0) perstar[(size_t)w * st.n + st.perm[slot]] = NEG_INF;
            }
        return;
    }
    // ---- once per workgroup: the mass column(s) in LDS (the seed's bracket search; SAMPLE's node masses).  The magnitude
    // rows are not staged: every per-node quantity the star loop needs comes from the call's table (k_marg_table).
    const double *lds_mass[NPOPS];
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        double *dst = smem + (size_t)kp * (mass_cap + 8);
        const double2 *src2 = reinterpret_cast<const double2 *>(iso_g[kp].mass);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        for (int j = tid; j < mass_cap / 2; j += 256) d2[j] = src2[j];                       // (mass_cap is even)
        if (tid < 8) dst[mass_cap + tid] = __builtin_inf();                                  // find_bracket's masked over-read
        lds_mass[kp] = dst;
    }
    __syncthreads();
    // Chunk-level pruning table, per 64-node chunk c of the primary-mass loop and per filter f (three planes):
    //   faint[c][f]  = the FAINTEST magnitude among the chunk's EEP rows.  Every node of the chunk interpolates
    //                  between those rows and a companion only adds flux, so no system of the chunk is fainter:
    //                  where even that is brighter than observed, every node pays the excess (chunk form of (A)).
    //   brt[c][f]    = the brightest a system of the chunk can be: brightest row of the chunk (primary) plus the
    //                  brightest row at or below the chunk (a companion is less massive than its primary, so its
    //                  two bracketing rows lie at or below the chunk's last row).  Where even that is fainter
    //                  than observed, every node and every mass ratio pays the deficit.
    //   (third plane: the chunk's own brightest row, an intermediate of the prefix minimum.)
    double *const chunk_tab = smem + (size_t)NPOPS * (mass_cap + 8);
    const size_t plane = (size_t)NPOPS * chunk_cap * NFP;
    if (chunk_cap > 0) {
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) {
            const int n = iso_g[kp].n, n_chunks = ((n - 1) * K + 63) >> 6;
            for (int idx = tid; idx < n_chunks * NFP; idx += 256) {
                const int c = idx / NFP, f = idx - c * NFP;
                const int r0 = (64 * c) / K;
                int r1 = (64 * c + 63) / K + 1;
                r1 = r1 > n - 1 ? n - 1 : r1;
                const double *col = iso_g[kp].mags + f;                          // (global, L2: once per workgroup)
                double mx = col[(size_t)r0 * NFP], mn = mx;
                for (int r = r0 + 1; r <= r1; ++r) { const double v = col[(size_t)r * NFP]; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
                chunk_tab[((size_t)kp * chunk_cap + c) * NFP + f] = mx;
                chunk_tab[2 * plane + ((size_t)kp * chunk_cap + c) * NFP + f] = mn;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) {
            const int n = iso_g[kp].n, n_chunks = ((n - 1) * K + 63) >> 6;
            for (int idx = tid; idx < n_chunks * NFP; idx += 256) {
                const int c = idx / NFP, f = idx - c * NFP;
                const double own = chunk_tab[2 * plane + ((size_t)kp * chunk_cap + c) * NFP + f];
                double pre = own;                              // brightest row at or below the chunk
                for (int cc = 0; cc < c; ++cc) { const double v = chunk_tab[2 * plane + ((size_t)kp * chunk_cap + cc) * NFP + f]; pre = v < pre ? v : pre; }
                // -2.5 log10(10^(-0.4 own) + 10^(-0.4 pre)), pre <= own:  pre - 2.5 log10(1 + 10^(-0.4 (own - pre)));
                // lowered by 1e-9 mag so that rounding can only make the bound weaker, never wrong
                chunk_tab[plane + ((size_t)kp * chunk_cap + c) * NFP + f] =
                    (pre - (2.5 / LN10) * log1p(exp((-0.4 * LN10) * (own - pre)))) - 1e-9;
            }
        }
    }
    // Upper bound of (log prior + log weight) over all nodes of a population: the IMF density per unit mass falls with
    // mass above 0.1 Msun, so its maximum is at the first point; the widest EEP interval bounds the weight.  Lets dead
    // nodes skip the logarithms of their own prior.  (Per walker and population: once per workgroup, by every wave.)
    double bmax[NPOPS];
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const double *mass = lds_mass[kp];
        double dmax = 0.0;
        for (int e2 = lane; e2 + 1 < iso_g[kp].n; e2 += 64) { const double dd = mass[e2 + 1] - mass[e2]; dmax = dd > dmax ? dd : dmax; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(dmax, o, 64); dmax = t > dmax ? t : dmax; }
        const double mlow = mass[0] > 0.1 ? mass[0] : 0.1;
        bmax[kp] = (dmax > 0.0) ? log_prior_mass_dev(pk.log_mass_norm, mlow) + log_pos(dmax / K / Q) : NEG_INF;
    }
    __syncthreads();                                       // the table; from here on every wave is on its own
    double *const wave_scr = chunk_tab + 3 * plane + (size_t)wave * B9_MARG_WAVE_SCRATCH(NFP);
    double *const s_shift = wave_scr, *const s_obs = wave_scr + NFP, *const s_wgt = wave_scr + 2 * NFP;
    int *const my_list = reinterpret_cast<int *>(chunk_tab + 3 * plane + (size_t)4 * B9_MARG_WAVE_SCRATCH(NFP)) + (size_t)wave * chunk_cap;
    const double mod = par[B9_P_MOD], av = par[B9_P_ABS];
    const int lf = lane & (NFP - 1), lslot = lane / NFP;   // pre-pass layout: this lane's filter and chunk slot
    constexpr int IPP = 64 / NFP;                          // chunks per pre-pass round

    for (int slot = wave0; slot < st.n_pad; slot += wave_stride) {
        const int orig = st.perm[slot];
        if (orig < 0) { if (lane == 0) vals[(size_t)w * st.n_pad + slot] = 0.0; continue; }
        // the star's per-filter constants -> this wave's LDS scratch (they are wave-uniform: kept in registers they cost
        // 6 NFP VGPRs of every lane)
        {
            double sh = 0.0;
#pragma unroll
            for (int f = 0; f < NFP; ++f) sh = (lane == f) ? mod + pk.abs_m1[f] * av : sh;
            if (lane < NFP) {
                s_shift[lane] = sh;
                s_obs[lane] = st.obs[B9_SIDX(NFP, lane, slot)];
                s_wgt[lane] = st.w[B9_SIDX(NFP, lane, slot)];
            }
        }
        const double c0m = st.c0m[slot], la = st.la[slot];
        if ((st.flags[slot] >> 8) == B9_STAGE_WD) continue;          // WD-stage stars: k_star_marg_wd (their own launch)
        wave_lds_fence();

        double ll[NPOPS];
        Best best; best.key = NEG_INF; best.mass = 0.0; best.ratio = 0.0; best.pop = 0;
        const unsigned long long g_row = SAMPLE ? (unsigned long long)(ms.row0 + w) : 0ull;
        double lw_pop[2] = {0.0, 0.0};                       // log weight of the population in the key
        if (SAMPLE && NPOPS == 2) { const double lam = par[B9_P_LAMBDA]; lw_pop[0] = log(lam); lw_pop[1] = log1p(-lam); }
        // one candidate node: term = its log-term, id = its index in the star's node list
#define B9_SAMPLE_NODE(term, id, m_, r_)                                                              \
        if (SAMPLE) {                                                                                     \
            const double key_ = (term) + lw_pop[kp] + gumbel(ms.k0, ms.k1, g_row, (unsigned)orig, (unsigned long long)(id), (unsigned)kp); \
            if (key_ > best.key) { best.key = key_; best.mass = (m_); best.ratio = (r_); best.pop = kp; }  \
        }
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) {
            const double *const mass = lds_mass[kp];
            // this (walker, population)'s table: the COMBINED magnitudes of (node, companion j) [(j - 1) NFP + f][npad], then per
            // node the primary's magnitudes [f][npad] and log(prior dM / Q) [npad]
            const double *const tab_wp = tab + (size_t)(w * NPOPS + kp) * tab_stride;
            const double *const tab_p1 = tab_wp + (size_t)(Q - 1) * NFP * npad, *const tab_base = tab_p1 + (size_t)NFP * npad;
            const int n_eep = iso_g[kp].n;
            const double tip = iso_g[kp].tip;
            Lse acc; acc.mx = NEG_INF; acc.sm = 0.0;
            {
                // Pruning (exact to ~1e-13 relative): a node whose log-term lies more than B9_MARG_CUT
                // below the wave's running maximum adds < e^-40 of the leading term and is dropped.
                //  (A) a companion only ADDS flux, so every filter in which the primary alone is already
                //      brighter than observed keeps at least that chi^2 for every mass ratio: a node whose
                //      lower bound is past the cut never enters the companion pass.
                const int n_nodes = (n_eep - 1) * K;
                // seed of the running maximum: the single-star term of the GRID NODE just below the star's
                // catalogue mass -- an actual term of the sum, hence a rigorous lower bound of its maximum
                // (only ever used as a pruning bound), so pruning bites from the first iteration
                double seed = NEG_INF;
                {
                    const double ms_ = st.mass1[slot];
                    if (ms_ >= mass[0] && ms_ <= tip) {
                        int lo; double t;
                        find_bracket(mass, n_eep, ms_, lo, t);
                        const double a = mass[lo], d = mass[lo + 1] - a;
                        if (d > 0.0) {
                            const double dMs = d / K;
                            int s = (int)((ms_ - a) / dMs);
                            s = s < 0 ? 0 : (s > K - 1 ? K - 1 : s);
                            const int ns = __builtin_amdgcn_readfirstlane(lo * K + s);       // the node: wave-uniform (scalar loads)
                            double c = 0.0;
#pragma unroll
                            for (int f = 0; f < NFP; ++f) { const double dd = (tab_p1[(size_t)f * npad + ns] + s_shift[f]) - s_obs[f]; c = fma(s_wgt[f] * dd, dd, c); }
                            const double bs = tab_base[ns];
                            if (isfinite(c) && bs != NEG_INF) seed = bs - 0.5 * c;
                        }
                    }
                }
                // Pre-pass over the chunk table with the lanes laid out as (chunk, filter): 64 / NFP chunks are bounded
                // per pass (one table word and one multiply-add per lane, a log2(NFP)-step shuffle sum), against
                // the SEED of the running maximum -- a looser cut than the loop's own test below uses, so the
                // survivors are a superset of the chunks that test keeps and the result is unchanged.  Their
                // indices, in ascending order, go to this wave's list in LDS.
                int n_list = -1;                                   // -1: no list, visit every chunk
                if (chunk_cap > 0) {
                    const int n_chunks = (n_nodes + 63) >> 6;
                    const double off = s_shift[lf] - s_obs[lf], wg = s_wgt[lf];
                    const double cut0 = 2.0 * ((bmax[kp] - seed) + B9_MARG_CUT);          // +inf without a seed: nothing is dropped here
                    n_list = 0;
                    for (int c0 = 0; c0 < n_chunks; c0 += IPP) {
                        const int c = c0 + lslot;
                        const bool in = c < n_chunks;
                        const double *cm = chunk_tab + ((size_t)kp * chunk_cap + (in ? c : 0)) * NFP;
                        const double too_bright = cm[lf] + off, too_faint = cm[plane + lf] + off;
                        const double dd = too_bright < 0.0 ? too_bright : (too_faint > 0.0 ? too_faint : 0.0);
                        double term = (wg * dd) * dd;
#pragma unroll
                        for (int o = NFP / 2; o > 0; o >>= 1) term += __shfl_xor(term, o, 64);
                        const bool keep = in && lf == 0 && !(term > cut0);
                        const unsigned long long m = __ballot(keep);
                        if (keep) my_list[n_list + __popcll(m & ((1ull << lane) - 1ull))] = c;
                        n_list += __popcll(m);
                    }
                    wave_lds_fence();
                }
                const int n_visit = n_list >= 0 ? n_list : (n_nodes + 63) >> 6;
                for (int iv = 0; iv < n_visit; ++iv) {
                    const int p0 = (n_list >= 0 ? my_list[iv] : iv) << 6;
                    const int pnode = p0 + lane;
                    // wave-wide running maximum (conservative for every lane)
                    const double wmx = wave_max_all(acc.mx > seed ? acc.mx : seed);
                    MSTAT(0, 1);
                    if (chunk_cap > 0) {       // the whole chunk at once (wave-uniform: every lane reads the same LDS words)
                        const double *cm = chunk_tab + ((size_t)kp * chunk_cap + (p0 >> 6)) * NFP;
                        double cb = 0.0;
#pragma unroll
                        for (int f = 0; f < NFP; ++f) {
                            const double off = s_shift[f] - s_obs[f];
                            const double too_bright = cm[f] + off, too_faint = cm[plane + f] + off;
                            const double dd = too_bright < 0.0 ? too_bright : (too_faint > 0.0 ? too_faint : 0.0);
                            cb = fma(s_wgt[f] * dd, dd, cb);
                        }
                        if (cb > 2.0 * ((bmax[kp] - wmx) + B9_MARG_CUT)) continue;
                    }
                    MSTAT(1, 1);
                    // ---- primary pass: lane = node.  Magnitudes, fluxes and log(prior dM / Q) of the node: table words ------
                    double p1[NFP];
#pragma unroll
                    for (int f = 0; f < NFP; ++f) p1[f] = tab_p1[(size_t)f * npad + pnode];
                    const double base_n = tab_base[pnode];                       // -inf: past the last node, or an empty EEP interval
                    bool live = base_n != NEG_INF;
                    double m1 = 0.0;
                    if (SAMPLE) {                                                // (only the draws report the node's mass)
                        const int pn = pnode < n_nodes ? pnode : 0, e = pn / K, sb = pn - e * K;
                        const double a = mass[e];
                        m1 = fma((double)sb, (mass[e + 1] - a) / K, a);
                    }
                    // j = 0 (single star) and the too-bright lower bound for j >= 1
                    double chi0 = 0.0, chi_lb = 0.0;
#pragma unroll
                    for (int f = 0; f < NFP; ++f) {
                        const double dd = (p1[f] + s_shift[f]) - s_obs[f];
                        const double wdd = s_wgt[f] * dd;
                        chi0 = fma(wdd, dd, chi0);
                        chi_lb = dd < 0.0 ? fma(wdd, dd, chi_lb) : chi_lb;
                    }
                    // with the bound bmax on this node's (prior + weight) nothing of it can matter: skip
                    const double cut_ub = 2.0 * ((bmax[kp] - wmx) + B9_MARG_CUT);
                    live = live && !(chi_lb > cut_ub);                           // chi0 >= chi_lb
                    if (__ballot(live) == 0ull) continue;
                    MSTAT(2, 1);
                    const double base = live ? base_n : NEG_INF;
                    if (live && isfinite(chi0)) {
                        lse_add(acc, base - 0.5 * chi0);
                        B9_SAMPLE_NODE(base - 0.5 * chi0, (long long)pnode * Q, m1, 0.0)
                    }
                    const double cut = 2.0 * ((base - wmx) + B9_MARG_CUT);       // chi^2 beyond this is negligible
                    const bool want = live && !(chi_lb > cut);
                    const unsigned long long wmask = __ballot(want);
                    if (wmask == 0ull || Q < 2) continue;                        // (A) for the whole wave
                    MSTAT(3, 1); MSTAT(6, __popcll(wmask));
                    // ---- companions: the combined magnitude of (node, mass ratio j, filter) does not depend on the star either:
                    // a table word.  What is left per star is the chi^2.
                    const double *tp = tab_wp + pnode;
                    {
                        // (no look-ahead of the next mass ratio's words: measured 2.39 -> 2.23 ms per call without -- its
                        //  registers cost more than the latency it hid; 16 filters go eight at a time)
                        constexpr int FB = NFP < 8 ? NFP : 8;
                        for (int j = 1; j < Q; ++j) {
                            double chi2 = want ? 0.0 : __builtin_inf();
                            MSTAT(4, 1);
#pragma unroll 1
                            for (int f0 = 0; f0 < NFP; f0 += FB) {
                                double C8[FB];
#pragma unroll
                                for (int f = 0; f < FB; ++f) C8[f] = tp[((size_t)(j - 1) * NFP + f0 + f) * npad];
#pragma unroll
                                for (int f = 0; f < FB; ++f) {
                                    const double dd = (C8[f] + s_shift[f0 + f]) - s_obs[f0 + f];
                                    chi2 = fma(s_wgt[f0 + f] * dd, dd, chi2);
                                }
                            }
                            if (want && isfinite(chi2) && chi2 <= cut) {
                                lse_add(acc, base - 0.5 * chi2);
                                B9_SAMPLE_NODE(base - 0.5 * chi2, (long long)pnode * Q + j, m1, (double)j / (double)Q)
                            }
                        }
                    }
                }
            }
            // wavefront combine of the 64 partial log-sum-exps: the wave's maximum first, then every lane's sum rescaled to it
            // ONCE and a plain shuffle sum (one exponential per lane instead of one per lane and shuffle step)
            const double wm = wave_max_all(acc.mx);
            // (lane 0's sum: the same pairs in the same order as the xor butterfly gave it; the star's value is lane 0's)
            const double ssum = wave_sum((acc.mx == NEG_INF) ? 0.0 : acc.sm * exp_fast(acc.mx - wm));
            ll[kp] = (wm == NEG_INF) ? NEG_INF : c0m + (wm + log(ssum));
        }
#undef B9_SAMPLE_NODE
        if (SAMPLE) {      // wave argmax of the keys (ties keep the lower lane)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                Best b; b.key = __shfl_down(best.key, o, 64); b.mass = __shfl_down(best.mass, o, 64);
                b.ratio = __shfl_down(best.ratio, o, 64); b.pop = __shfl_down(best.pop, o, 64);
                if (b.key > best.key) best = b;
            }
        }
        if (lane == 0) {
            double l = ll[0];
            if (NPOPS == 2) { const double lam = par[B9_P_LAMBDA]; l = logaddexp(log(lam) + ll[0], log1p(-lam) + ll[NPOPS - 1]); }
            const double v = logaddexp(la, l);
            vals[(size_t)w * st.n_pad + slot] = v;
            if (perstar) perstar[(size_t)w * st.n + orig] = v;
            if (SAMPLE) {
                const size_t o = (size_t)w * st.n + orig;
                const bool any = best.key != NEG_INF;
                ms.mass[o] = any ? best.mass : 0.0;
                ms.ratio[o] = any ? best.ratio : 0.0;
                ms.member[o] = (l == NEG_INF) ? 0.0 : exp(l - v);       // p L_cluster / (p L_cluster + (1 - p) L_field)
                if (ms.pop) ms.pop[o] = any ? best.pop : 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_marg_table: one call's node table.  For walker-population wp, primary node n (EEP interval n / K,
// sub-step n % K: the primary mass the main kernel forms, same operations) and mass ratio j / Q, j = 1 .. Q-1:
// secondary mass m2 = (j / Q) m1, bracket + linear interpolation of the derived isochrone's rows, and the SYSTEM's combined
// magnitude per filter.  And per node the primary itself: magnitudes, log(prior(m1) dM / Q) (-inf: no such node).
// Layout tab[wp]: [(j - 1) * NFP + f][npad] combined magnitudes, [f][npad] primary magnitudes, [npad] log weights
// (npad = whole 64-node chunks; nodes past the end hold 0 / -inf).
// Grid: (walkers * pops, parts).
// ------------------------------------------------------------------------------------------
template <int NFP>
__global__ __launch_bounds__(256) void k_marg_table(const IsoHdr *__restrict__ hdr, const double *__restrict__ iso_data, long long iso_stride,
                                                    int mass_cap, int K, int Q, double *__restrict__ tab, long long tab_stride, int npad, double lmn)
{
    extern __shared__ __attribute__((aligned(16))) double s_mass[];
    const int wp = blockIdx.x, tid = threadIdx.x;
    const IsoHdr h = hdr[wp];
    if (!h.valid) return;
    const double *g_mass = iso_data + (size_t)wp * iso_stride, *g_mags = g_mass + mass_cap;
    for (int j = tid; j < mass_cap; j += 256) s_mass[j] = g_mass[j];
    if (tid < 8) s_mass[mass_cap + tid] = __builtin_inf();              // find_bracket's masked over-read
    __syncthreads();
    const int n_nodes = (h.n - 1) * K;
    double *out = tab + (size_t)wp * tab_stride;
    // Every (node, mass ratio j >= 1): the system's combined magnitude per filter,
    //     -2.5 log10(10^(-0.4 p1) + 10^(-0.4 p2)),   p1 / p2 = the primary's / companion's interpolated magnitudes
    // (companion below the isochrone's first point: no flux, magnitude 99.999).
    for (int idx = blockIdx.y * 256 + tid; idx < npad * (Q - 1); idx += gridDim.y * 256) {
        const int jm1 = idx / npad, node = idx - jm1 * npad, j = jm1 + 1;
        double C[NFP];
#pragma unroll
        for (int f = 0; f < NFP; ++f) C[f] = 0.0;
        if (node < n_nodes) {
            const int e = node / K, s = node - e * K;
            const double a = s_mass[e], d = s_mass[e + 1] - a;
            if (d > 0.0) {
                const double dM = d / K;
                const double m1 = fma((double)s, dM, a);
                const double t1 = (m1 - a) / d;
                const double *r0 = g_mags + (size_t)e * NFP;
                const double m2 = ((double)j / (double)Q) * m1;
                const bool dark2 = m2 < s_mass[0];
                int lo2; double t2;
                find_bracket(s_mass, h.n, m2, lo2, t2);
                const double *r2 = g_mags + (size_t)lo2 * NFP;
#pragma unroll
                for (int f = 0; f < NFP; ++f) {
                    const double p1 = lerp(r0[f], r0[NFP + f], t1);
                    const double p2 = dark2 ? B9_MAG_NOFLUX : lerp(r2[f], r2[NFP + f], t2);
                    C[f] = (-2.5 / LN10) * log_pos(exp_fast((-0.4 * LN10) * p1) + exp_fast((-0.4 * LN10) * p2));
                }
            }
        }
#pragma unroll
        for (int f = 0; f < NFP; ++f) out[((size_t)jm1 * NFP + f) * npad + node] = C[f];
    }
    // the primaries: magnitudes and log(prior(m1) dM / Q) of every node
    double *p1o = out + (size_t)(Q - 1) * NFP * npad, *bs = p1o + (size_t)NFP * npad;
    for (int node = blockIdx.y * 256 + tid; node < npad; node += gridDim.y * 256) {
        double P[NFP], base = NEG_INF;
#pragma unroll
        for (int f = 0; f < NFP; ++f) P[f] = 0.0;
        if (node < n_nodes) {
            const int e = node / K, s = node - e * K;
            const double a = s_mass[e], d = s_mass[e + 1] - a;
            const double *r0 = g_mags + (size_t)e * NFP;
            const double dM = d / K;
            const double m1 = fma((double)s, dM, a);
            const double t1 = (d > 0.0) ? (m1 - a) / d : 0.0;
#pragma unroll
            for (int f = 0; f < NFP; ++f) P[f] = lerp(r0[f], r0[NFP + f], t1);
            if (d > 0.0) base = log_prior_mass_dev(lmn, m1) + log_pos(dM / Q);
        }
#pragma unroll
        for (int f = 0; f < NFP; ++f) p1o[(size_t)f * npad + node] = P[f];
        bs[node] = base;
    }
}

// ------------------------------------------------------------------------------------------
// k_star_marg_wd: the WD-stage stars of the marginalised mode, one wavefront per (walker, star): lanes stride over
// the 8 iso_increm primary-mass steps in (AGB tip, M_wd_up], each through the general WD branch (IFMR -> cooling
// -> atmosphere); online log-sum-exp per lane, wavefront-shuffle merge.  Grid: (ceil(n_wd / 4), walkers).
// ------------------------------------------------------------------------------------------
template <int NFP, int NPOPS, bool SAMPLE>
__global__ __launch_bounds__(256) void k_star_marg_wd(DevPack pk, DevStars st, const IsoHdr *__restrict__ hdr,
                                                      const double *__restrict__ iso_data, long long iso_stride,
                                                      int mass_cap, const double *__restrict__ params,
                                                      double *__restrict__ vals, double *__restrict__ perstar,
                                                      int K, MargSample ms)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), w = blockIdx.y;
    const int k_wd = blockIdx.x * 4 + wave;
    if (k_wd >= st.n_wd) return;
    const int slot = st.wd_slot[k_wd], orig = st.perm[slot];
    const double *par = params + (size_t)w * B9_NPARAM;
    IsoView<NFP> iso[NPOPS];
    double tip_min;
    const bool valid = load_iso_views<NFP, NPOPS>(hdr, iso_data, iso_stride, mass_cap, w, iso, tip_min);
    if (!valid) {
        if (lane == 0) { vals[(size_t)w * st.n_pad + slot] = 0.0; if (perstar) perstar[(size_t)w * st.n + orig] = NEG_INF; }
        return;
    }
    double obs[NFP], wgt[NFP], shift[NFP];
    const double mod = par[B9_P_MOD], av = par[B9_P_ABS];
#pragma unroll
    for (int f = 0; f < NFP; ++f) { obs[f] = st.obs[B9_SIDX(NFP, f, slot)]; wgt[f] = st.w[B9_SIDX(NFP, f, slot)]; shift[f] = mod + pk.abs_m1[f] * av; }
    const double c0m = st.c0m[slot], la = st.la[slot];
    const int wd_type = st.flags[slot] & 1;
    double ll[NPOPS];
    Best best; best.key = NEG_INF; best.mass = 0.0; best.ratio = 0.0; best.pop = 0;
    const unsigned long long g_row = SAMPLE ? (unsigned long long)(ms.row0 + w) : 0ull;
    double lw_pop[2] = {0.0, 0.0};
    if (SAMPLE && NPOPS == 2) { const double lam = par[B9_P_LAMBDA]; lw_pop[0] = log(lam); lw_pop[1] = log1p(-lam); }
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const IsoView<NFP> &is = iso[kp];
        Lse acc; acc.mx = NEG_INF; acc.sm = 0.0;
        WdAxes ax;
        ax.log_age = pk.log_age;
        const int ny = pk.n_y > 1 ? 2 : 1;
        for (int df = 0; df < 2; ++df) for (int dy = 0; dy < 2; ++dy)
            ax.tips[df * 2 + dy] = pk.tips + (size_t)((is.i_feh + df) * pk.n_y + (is.i_y + (dy < ny ? dy : 0))) * pk.n_age;
        ax.wc_log_age_lds = nullptr; ax.wc_track = pk.wc_track; ax.wc_mass = pk.wc_mass; ax.wc_carb = pk.wc_carb;
        ax.at_log_teff = pk.at_log_teff; ax.at_logg = pk.at_logg;
        const int steps = 8 * K;
        const double dM = (pk.m_wd_up - is.tip) / steps;
        if (dM > 0.0) {
            const double log_w = log(dM);
            for (int j = 1 + lane; j <= steps; j += 64) {
                const double m1 = is.tip + dM * j;
                double p[NFP];
                star_mags<NFP>(pk, ax, is, par, m1, wd_type, p);
                double chi2 = 0.0;
#pragma unroll
                for (int f = 0; f < NFP; ++f) { const double d = (p[f] + shift[f]) - obs[f]; chi2 = fma(wgt[f] * d, d, chi2); }
                if (isfinite(chi2)) {
                    const double term = (log_prior_mass_dev(pk.log_mass_norm, m1) - 0.5 * chi2) + log_w;
                    lse_add(acc, term);
                    if (SAMPLE) {
                        const double key_ = term + lw_pop[kp] + gumbel(ms.k0, ms.k1, g_row, (unsigned)orig, (unsigned long long)j, (unsigned)kp);
                        if (key_ > best.key) { best.key = key_; best.mass = m1; best.ratio = 0.0; best.pop = kp; }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Lse b; b.mx = __shfl_down(acc.mx, o, 64); b.sm = __shfl_down(acc.sm, o, 64);
            acc = lse_merge(acc, b);
        }
        ll[kp] = (acc.mx == NEG_INF) ? NEG_INF : c0m + (acc.mx + log(acc.sm));
    }
    if (SAMPLE) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Best b; b.key = __shfl_down(best.key, o, 64); b.mass = __shfl_down(best.mass, o, 64);
            b.ratio = __shfl_down(best.ratio, o, 64); b.pop = __shfl_down(best.pop, o, 64);
            if (b.key > best.key) best = b;
        }
    }
    if (lane == 0) {
        double l = ll[0];
        if (NPOPS == 2) { const double lam = par[B9_P_LAMBDA]; l = logaddexp(log(lam) + ll[0], log1p(-lam) + ll[NPOPS - 1]); }
        const double v = logaddexp(la, l);
        vals[(size_t)w * st.n_pad + slot] = v;
        if (perstar) perstar[(size_t)w * st.n + orig] = v;
        if (SAMPLE) {
            const size_t o = (size_t)w * st.n + orig;
            const bool any = best.key != NEG_INF;
            ms.mass[o] = any ? best.mass : 0.0;
            ms.ratio[o] = any ? best.ratio : 0.0;
            ms.member[o] = (l == NEG_INF) ? 0.0 : exp(l - v);
            if (ms.pop) ms.pop[o] = any ? best.pop : 0;
        }
    }
}
