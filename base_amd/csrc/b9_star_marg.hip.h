// b9_star_marg.hip.h -- k_star_marg: marginalised mode, one wavefront per star (+ the sampleMass draws).
// Part of the single translation unit b9_kernels.hip (included there, in this order); gfx950 only.
#pragma once

// ------------------------------------------------------------------------------------------
// k_star_marg  (marginalised mode; SURVEY 8a row a6 "marg.cpp-like", [RECALL] margEvolveWithBinary)
//
// ONE WAVEFRONT PER STAR.  The star's likelihood is integrated over primary mass (iso_increm equal
// sub-steps inside every EEP interval of the derived isochrone, left-endpoint rule) and mass ratio
// (n_q nodes j / n_q): lane l takes primary nodes l, l + 64, ...; for each it interpolates the
// primary once and loops over the mass ratios (secondary: binary search in the LDS-resident mass
// column, rows from LDS, flux combine); every node contributes exp(ll) dM / n_q to a per-lane
// online log-sum-exp, the 64 lanes are combined with wavefront shuffles, and the star's value is
// written to its slot (the per-walker sum over stars is k_finalize's fixed-order block sum).
// A star of stage WD integrates over (AGB tip, M_wd_up] in 8 iso_increm steps through the WD branch.
// The whole isochrone (mass column + magnitude rows) of the walker lives in LDS.
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

__device__ __forceinline__ double log_prior_mass_dev(double lmn, double m)
{
    const double z = (log10(m) - MF_MU) / MF_SIGMA;
    return lmn - 0.5 * z * z - log(m) - log(LN10);
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

#ifndef B9_MARG_MIN_WAVES
#define B9_MARG_MIN_WAVES 3      // measured on 50k x 8 x 8, 6384 nodes: 2 waves/SIMD (209 VGPRs) 3.24e7 star-evals/s, 3 waves (168, the WD branch spills) 3.87e7, 4 waves 2.50e7
#endif
template <int NFP, int NPOPS, bool SAMPLE>
__global__ __launch_bounds__(256, B9_MARG_MIN_WAVES) void k_star_marg(DevPack pk, DevStars st, const IsoHdr *__restrict__ hdr,
                                                    const double *__restrict__ iso_data, long long iso_stride,
                                                    int mass_cap, const double *__restrict__ params,
                                                    double *__restrict__ vals, double *__restrict__ perstar,
                                                    int K, int Q, MargSample ms, int chunk_cap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), w = blockIdx.y;
    const double *par = params + (size_t)w * B9_NPARAM;
    IsoView<NFP> iso[NPOPS];
    double tip_min;
    const bool valid = load_iso_views<NFP, NPOPS>(hdr, iso_data, iso_stride, mass_cap, w, iso, tip_min);
    const int slot = blockIdx.x * 4 + wave;
    if (!valid) {
        if (slot < st.n_pad && lane == 0) {
            vals[(size_t)w * st.n_pad + slot] = 0.0;
            if (perstar && st.perm[slot] >= 0) perstar[(size_t)w * st.n + st.perm[slot]] = NEG_INF;
        }
        return;
    }
    // stage the isochrone(s): [pop][ mass[cap] | mags[cap][NFP] ]
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        double *dst = smem + (size_t)kp * mass_cap * (NFP + 1);
        const double2 *src = reinterpret_cast<const double2 *>(iso[kp].mass);
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        const int n2 = (mass_cap + iso[kp].n * NFP + 1) / 2;         // mass column (full capacity) + n rows
        for (int j = tid; j < n2; j += 256) d2[j] = src[j];
        iso[kp].mass = dst; iso[kp].mags = dst + mass_cap;
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
    double *const chunk_tab = smem + (size_t)NPOPS * mass_cap * (NFP + 1);
    const size_t plane = (size_t)NPOPS * chunk_cap * NFP;
    if (chunk_cap > 0) {
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) {
            const int n = iso[kp].n, n_chunks = ((n - 1) * K + 63) >> 6;
            for (int idx = tid; idx < n_chunks * NFP; idx += 256) {
                const int c = idx / NFP, f = idx - c * NFP;
                const int r0 = (64 * c) / K;
                int r1 = (64 * c + 63) / K + 1;
                r1 = r1 > n - 1 ? n - 1 : r1;
                double mx = iso[kp].mags[(size_t)r0 * NFP + f], mn = mx;
                for (int r = r0 + 1; r <= r1; ++r) { const double v = iso[kp].mags[(size_t)r * NFP + f]; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
                chunk_tab[((size_t)kp * chunk_cap + c) * NFP + f] = mx;
                chunk_tab[2 * plane + ((size_t)kp * chunk_cap + c) * NFP + f] = mn;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) {
            const int n = iso[kp].n, n_chunks = ((n - 1) * K + 63) >> 6;
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
        __syncthreads();
    }
    if (slot >= st.n_pad) return;
    const int orig = st.perm[slot];
    if (orig < 0) { if (lane == 0) vals[(size_t)w * st.n_pad + slot] = 0.0; return; }

    double obs[NFP], wgt[NFP];
#pragma unroll
    for (int f = 0; f < NFP; ++f) { obs[f] = st.obs[(size_t)f * st.n_pad + slot]; wgt[f] = st.w[(size_t)f * st.n_pad + slot]; }
    const double c0m = st.c0m[slot], la = st.la[slot];
    const int flags = st.flags[slot], stage = flags >> 8, wd_type = flags & 1;
    const double mod = par[B9_P_MOD], av = par[B9_P_ABS];
    double shift[NFP];
#pragma unroll
    for (int f = 0; f < NFP; ++f) shift[f] = mod + pk.abs_m1[f] * av;

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
        const IsoView<NFP> &is = iso[kp];
        Lse acc; acc.mx = NEG_INF; acc.sm = 0.0;
        if (stage == B9_STAGE_WD) {
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
                        B9_SAMPLE_NODE(term, j, m1, 0.0)
                    }
                }
            }
        } else {
            // Pruning (exact to ~1e-13 relative): a node whose log-term lies more than B9_MARG_CUT
            // below the wave's running maximum adds < e^-40 of the leading term and is dropped.
            //  (A) a companion only ADDS flux, so every filter in which the primary alone is already
            //      brighter than observed keeps at least that chi^2 for every mass ratio: if that lower
            //      bound is past the cut, the whole mass-ratio loop of this primary is skipped;
            //  (B) inside a node the filters are accumulated one at a time and the wave leaves the
            //      filter loop as soon as EVERY lane's partial chi^2 is past the cut.
            const int n_nodes = (is.n - 1) * K;
            // seed of the running maximum: the single-star term of the GRID NODE just below the star's
            // catalogue mass -- an actual term of the sum, hence a rigorous lower bound of its maximum
            // (only ever used as a pruning bound), so pruning bites from the first iteration
            double seed = NEG_INF;
            {
                const double ms = st.mass1[slot];
                if (ms >= is.mass[0] && ms <= is.tip) {
                    int lo; double t;
                    find_bracket(is.mass, is.n, ms, lo, t);
                    const double a = is.mass[lo], d = is.mass[lo + 1] - a;
                    if (d > 0.0) {
                        const double dMs = d / K;
                        int s = (int)((ms - a) / dMs);
                        s = s < 0 ? 0 : (s > K - 1 ? K - 1 : s);
                        const double mn = fma((double)s, dMs, a), tn = (mn - a) / d;
                        const double *r = is.mags + (size_t)lo * NFP;
                        double c = 0.0;
#pragma unroll
                        for (int f = 0; f < NFP; ++f) { const double dd = (lerp(r[f], r[NFP + f], tn) + shift[f]) - obs[f]; c = fma(wgt[f] * dd, dd, c); }
                        if (isfinite(c)) seed = (log_prior_mass_dev(pk.log_mass_norm, mn) + log(dMs / Q)) - 0.5 * c;
                    }
                }
            }
            // upper bound of (log prior + log weight) over all nodes: the IMF density per unit mass
            // falls with mass above 0.1 Msun, so its maximum is at the first point; the widest EEP
            // interval bounds the weight.  Lets dead nodes skip the two logarithms of their own prior.
            double dmax = 0.0;
            for (int e2 = lane; e2 + 1 < is.n; e2 += 64) { const double dd = is.mass[e2 + 1] - is.mass[e2]; dmax = dd > dmax ? dd : dmax; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(dmax, o, 64); dmax = t > dmax ? t : dmax; }
            const double mlow = is.mass[0] > 0.1 ? is.mass[0] : 0.1;
            const double bmax = (dmax > 0.0) ? log_prior_mass_dev(pk.log_mass_norm, mlow) + log(dmax / K / Q) : NEG_INF;
            // Pre-pass over the chunk table with the lanes laid out as (chunk, filter): 64 / NFP chunks are bounded
            // per pass (one table word and one multiply-add per lane, a log2(NFP)-step shuffle sum), against
            // the SEED of the running maximum -- a looser cut than the loop's own test below uses, so the
            // survivors are a superset of the chunks that test keeps and the result is unchanged.  Their
            // indices, in ascending order, go to this wave's list in LDS.
            int n_list = -1;                                   // -1: no list, visit every chunk
            int *const my_list = reinterpret_cast<int *>(chunk_tab + 3 * plane + (size_t)4 * 2 * NFP) + (size_t)wave * chunk_cap;
            if (chunk_cap > 0) {
                double *const pre = chunk_tab + 3 * plane + (size_t)wave * 2 * NFP;
                if (lane == 0) {
#pragma unroll
                    for (int f = 0; f < NFP; ++f) { pre[f] = shift[f] - obs[f]; pre[NFP + f] = wgt[f]; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int f = lane & (NFP - 1), cg = lane / NFP, n_chunks = (n_nodes + 63) >> 6;
                const double off = pre[f], wg = pre[NFP + f];
                const double cut0 = 2.0 * ((bmax - seed) + B9_MARG_CUT);          // +inf without a seed: nothing is dropped here
                n_list = 0;
                for (int c0 = 0; c0 < n_chunks; c0 += 64 / NFP) {
                    const int c = c0 + cg;
                    const bool in = c < n_chunks;
                    const double *cm = chunk_tab + ((size_t)kp * chunk_cap + (in ? c : 0)) * NFP;
                    const double too_bright = cm[f] + off, too_faint = cm[plane + f] + off;
                    const double dd = too_bright < 0.0 ? too_bright : (too_faint > 0.0 ? too_faint : 0.0);
                    double term = (wg * dd) * dd;
#pragma unroll
                    for (int o = NFP / 2; o > 0; o >>= 1) term += __shfl_xor(term, o, 64);
                    const bool keep = in && f == 0 && !(term > cut0);
                    const unsigned long long m = __ballot(keep);
                    if (keep) my_list[n_list + __popcll(m & ((1ull << lane) - 1ull))] = c;
                    n_list += __popcll(m);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            const int n_visit = n_list >= 0 ? n_list : (n_nodes + 63) >> 6;
            for (int iv = 0; iv < n_visit; ++iv) {
                const int p0 = (n_list >= 0 ? my_list[iv] : iv) << 6;
                const int pnode = p0 + lane;
                // wave-wide running maximum (conservative for every lane)
                double wmx = acc.mx > seed ? acc.mx : seed;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(wmx, o, 64); wmx = t > wmx ? t : wmx; }
                MSTAT(0, 1);
                if (chunk_cap > 0) {       // the whole chunk at once (wave-uniform: every lane reads the same LDS words)
                    const double *cm = chunk_tab + ((size_t)kp * chunk_cap + (p0 >> 6)) * NFP;
                    double cb = 0.0;
#pragma unroll
                    for (int f = 0; f < NFP; ++f) {
                        const double off = shift[f] - obs[f];
                        const double too_bright = cm[f] + off, too_faint = cm[plane + f] + off;
                        const double dd = too_bright < 0.0 ? too_bright : (too_faint > 0.0 ? too_faint : 0.0);
                        cb = fma(wgt[f] * dd, dd, cb);
                    }
                    if (cb > 2.0 * ((bmax - wmx) + B9_MARG_CUT)) continue;
                }
                MSTAT(1, 1);
                bool live = pnode < n_nodes;
                int e = 0, s = 0;
                double a = 0.0, d = 1.0;
                if (live) { e = pnode / K; s = pnode - e * K; a = is.mass[e]; d = is.mass[e + 1] - a; live = d > 0.0; }
                const double dM = d / K;
                const double m1 = fma((double)s, dM, a);
                const double t1 = (m1 - a) / d;
                double p1[NFP];
                const double *r0 = is.mags + (size_t)e * NFP;
#pragma unroll
                for (int f = 0; f < NFP; ++f) p1[f] = lerp(r0[f], r0[NFP + f], t1);
                // j = 0 (single star) and the too-bright lower bound for j >= 1
                double chi0 = 0.0, chi_lb = 0.0;
#pragma unroll
                for (int f = 0; f < NFP; ++f) {
                    const double dd = (p1[f] + shift[f]) - obs[f];
                    chi0 = fma(wgt[f] * dd, dd, chi0);
                    chi_lb = dd < 0.0 ? fma(wgt[f] * dd, dd, chi_lb) : chi_lb;
                }
                // with the bound bmax on this node's (prior + weight) nothing of it can matter: skip
                const double cut_ub = 2.0 * ((bmax - wmx) + B9_MARG_CUT);
                live = live && !(chi_lb > cut_ub);                           // chi0 >= chi_lb
                if (__ballot(live) == 0ull) continue;
                MSTAT(2, 1);
                const double base = live ? log_prior_mass_dev(pk.log_mass_norm, m1) + log(dM / Q) : NEG_INF;
                if (live && isfinite(chi0)) {
                    lse_add(acc, base - 0.5 * chi0);
                    B9_SAMPLE_NODE(base - 0.5 * chi0, (long long)pnode * Q, m1, 0.0)
                }
                const double cut = 2.0 * ((base - wmx) + B9_MARG_CUT);       // chi^2 beyond this is negligible
                bool want = live && !(chi_lb > cut);
                if (__ballot(want) == 0ull) continue;                        // (A) for the whole wave
                MSTAT(3, 1); MSTAT(6, __popcll(__ballot(want)));
                for (int j = 1; j < Q; ++j) {
                    const double m2 = ((double)j / (double)Q) * m1;
                    const bool dark2 = m2 < is.mass[0];
                    int lo2; double t2;
                    find_bracket(is.mass, is.n, m2, lo2, t2);
                    const double *s0 = is.mags + (size_t)lo2 * NFP;
                    double chi2 = want ? 0.0 : __builtin_inf();
                    bool done = false;
                    MSTAT(4, 1);
#pragma unroll
                    for (int f = 0; f < NFP; ++f) {
                        if (!done) {
                            MSTAT(5, 1);
                            const double p2 = dark2 ? B9_MAG_NOFLUX : lerp(s0[f], s0[NFP + f], t2);
                            const double pc = p1[f] - (2.5 / LN10) * log1pexp((-0.4 * LN10) * (p2 - p1[f]));
                            const double dd = (pc + shift[f]) - obs[f];
                            chi2 = fma(wgt[f] * dd, dd, chi2);
                            done = (__ballot(chi2 <= cut) == 0ull);          // (B): uniform across the wave
                        }
                    }
                    if (want && !done && isfinite(chi2) && chi2 <= cut) {
                        lse_add(acc, base - 0.5 * chi2);
                        B9_SAMPLE_NODE(base - 0.5 * chi2, (long long)pnode * Q + j, m1, (double)j / (double)Q)
                    }
                }
            }
        }
        // wavefront shuffle reduction of the 64 partial log-sum-exps
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Lse b; b.mx = __shfl_down(acc.mx, o, 64); b.sm = __shfl_down(acc.sm, o, 64);
            acc = lse_merge(acc, b);
        }
        ll[kp] = (acc.mx == NEG_INF) ? NEG_INF : c0m + (acc.mx + log(acc.sm));
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

