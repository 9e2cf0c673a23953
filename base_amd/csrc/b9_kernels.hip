// b9_kernels.hip -- hand-written gfx950 kernels of the BASE-9 per-step log-posterior path.
//
//   k_derive_iso   SURVEY 8a rows a3, a8  one workgroup per (walker, population, slice of EEPs): grid
//                                     brackets by wave ballot, EEP-range intersection, EEP-wise
//                                     tri-linear interpolation (one value per thread).  In the
//                                     device-resident sampler it first FINISHES the previous step
//                                     (fixed-order sum of partials + prior + Metropolis accept,
//                                     recomputed identically by every workgroup of the walker) and
//                                     draws the new proposal (Philox4x32-10 + Box-Muller)
//   k_star_like    rows a4-a7, a9      given-mass mode, the dominant kernel.  HOT workgroups: one
//                                     LANE per star -- 8-ary search in the LDS-staged mass column,
//                                     magnitude rows from L2, binary flux combination, Gaussian
//                                     chi^2, population mixture, product-form field-star mixture,
//                                     one partial per wave.  HEAVY workgroups (first in the grid):
//                                     the stars above the AGB tip (IFMR -> WD cooling -> WD
//                                     atmosphere, or NS/BH) through the general per-star code
//   k_star_marg    row a6 (marg.)      marginalised mode: ONE WAVEFRONT PER STAR integrating over
//                                     primary mass and mass ratio, rigorous pruning, online
//                                     log-sum-exp, wavefront-shuffle merge
//   k_finalize     row a8              fixed-order sum of the partials + cluster prior (+ the accept
//                                     of a sampler block's last step)
//
// One MCMC step = k_derive_iso + k_star_like (two launches, no host involvement).
//
// The reference source is not mounted (/root/reference/README.md:4), so none of this can
// cite a reference file:line; DESIGN.md "Math" is the normative restatement and
// oracle/b9_oracle.c the CPU checker.  Floating-point contract: built with
// -ffp-contract=off; every interpolation is an explicit fma(t, b - a, a), which makes the
// derived isochrone bit-identical to the oracle's.
//
// No MFMA anywhere: there is no dense contraction on this path (BASELINE.json north_star).
// Diagnostic-only macros (never defined in the shipped library): B9_STAMPS (per-phase s_memtime
// stamps), B9_ABL_* (ablation builds used for the attribution in DESIGN.md section 8).
#include "b9_device.h"
#include "b9_launch.h"
#include <algorithm>
#include "../../include/base9_hip.h"

#define LOG_G_PLUS_LOG_MSUN 26.12302173752
#define MF_MU (-1.02)
#define MF_SIGMA 0.67729
#define LN10 2.302585092994045684
#define NEG_INF (-__builtin_inf())

// Diagnostic build only (-DB9_STAMPS): per-wave s_memtime stamps of the hot kernel's phases,
// written to a buffer of their own that no kernel reads.  Never defined in the shipped library.
#ifdef B9_STAMPS
#define B9_NSTAMP 12
__device__ unsigned long long g_stamps[8192 * B9_NSTAMP];
#ifndef B9_STAMP_MASK
#define B9_STAMP_MASK 0xFFF       // which stamps are live (bit k); the rest compile to nothing
#endif
#define STAMP(k)                                                                                   \
    if ((B9_STAMP_MASK >> (k)) & 1)                                                                \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        const unsigned wv_ = blockIdx.x * 4 + (threadIdx.x >> 6);                                  \
        if ((threadIdx.x & 63) == 0 && wv_ < 8192) g_stamps[wv_ * B9_NSTAMP + (k)] = t_;           \
    } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

__device__ __forceinline__ double lerp(double a, double b, double t) { return fma(t, b - a, a); }

// largest i in [0, n-2] with ax[i] <= x (clamped); identical to the oracle's bracket()
__device__ __forceinline__ int bracket(const double *__restrict__ ax, int n, double x)
{
    int lo = 0, hi = n - 1;
    if (n < 2) return 0;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ax[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// log(x) for x >= 1 (also +inf / NaN in, NaN out).  The hot kernel only ever needs log(1 + r)
// with r >= 0, to an ABSOLUTE accuracy of a few 1e-16 -- so the argument reduction and
// polynomial of fdlibm's e_log.c (error < 1 ulp) are enough and the double-double arithmetic,
// subnormal and sign handling of the library log/log1p (98 / 135 VALU instructions each, and 9
// inlined copies per binary star) are not.  ~35 instructions; the one division is a v_rcp_f64
// seed plus two Newton steps and a residual correction.
__device__ __forceinline__ double log_ge1(double x)
{
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    int k = __builtin_amdgcn_frexp_exp(x);          // x = m * 2^k, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lt = m < 0.70710678118654752440;
    m = lt ? m + m : m;                             // m in [sqrt(1/2), sqrt(2))
    k = lt ? k - 1 : k;
    const double f = m - 1.0;
    const double y = 2.0 + f;
    double r = __builtin_amdgcn_rcp(y);
    r = fma(fma(-y, r, 1.0), r, r);
    r = fma(fma(-y, r, 1.0), r, r);
    double sq = f * r;
    sq = fma(fma(-y, sq, f), r, sq);                // s = f / (2 + f)
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return fma(dk, ln2_hi, -((hfsq - fma(sq, hfsq + R, dk * ln2_lo)) - f));
}

// exp(x) to 1 ulp for the hot kernel (~20 VALU instructions; the library exp is 42): Cody-Waite
// reduction by ln 2 and a degree-13 Horner polynomial on [-ln2/2, ln2/2], scaled by v_ldexp_f64.
// x is clamped to [-750, 750] (0 / +inf result); NaN propagates.
__device__ __forceinline__ double exp_fast(double x)
{
    x = x < -750.0 ? -750.0 : (x > 750.0 ? 750.0 : x);
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0); p = fma(p, r, 1.0 / 39916800.0); p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);    p = fma(p, r, 1.0 / 40320.0);    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);       p = fma(p, r, 1.0 / 120.0);      p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);         p = fma(p, r, 0.5);              p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// log(1 + exp(x)), any x (x = -inf gives 0)
__device__ __forceinline__ double log1pexp(double x) { return log_ge1(1.0 + exp_fast(x)); }

__device__ __forceinline__ double logaddexp(double a, double b)
{
    if (a == NEG_INF) return b;
    if (b == NEG_INF) return a;
    double hi = a > b ? a : b, lo = a > b ? b : a;
    return hi + log1pexp(lo - hi);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;   // valid in lane 0
}

// ------------------------------------------------------------------------------------------
// Counter-based random numbers for the device-resident sampler: Philox4x32-10 (Salmon et al.
// 2011), counter = (step lo, step hi, walker, draw), key = seed.  base_amd/mcmc.py holds the
// numpy twin; tests/test_mcmc.py checks it against the Random123 known-answer vectors.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                           unsigned k0, unsigned k1, unsigned (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)c0 * 0xD2511F53ull;
        const unsigned long long p1 = (unsigned long long)c2 * 0xCD9E8D57ull;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u01(unsigned hi, unsigned lo)
{
    const unsigned long long x = ((unsigned long long)(hi >> 5) << 26) + (unsigned long long)(lo >> 6);
    return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------------------------------
// Finishing a log-posterior evaluation: fixed-order sum of a walker's partials + cluster prior
// (SURVEY 8a row a8), and -- for the device-resident sampler -- the Metropolis accept/reject.
// Used by k_finalize (one workgroup per walker) and, redundantly by every workgroup of a walker,
// as the prologue of the NEXT step's k_derive_iso.
// ------------------------------------------------------------------------------------------
__device__ inline double log_prior_cluster(const DevPriors &pr, const double *__restrict__ par, int n_pops)
{
    if (!(par[B9_P_LOGAGE] >= pr.log_age_min && par[B9_P_LOGAGE] <= pr.log_age_max)) return NEG_INF;
    if (par[B9_P_ABS] < 0.0) return NEG_INF;
    if (n_pops == 2 && !(par[B9_P_LAMBDA] >= 0.0 && par[B9_P_LAMBDA] <= 1.0)) return NEG_INF;
    double lp = 0.0;
    for (int k = 0; k < B9_NPARAM; ++k) {
        if (k == B9_P_LOGAGE) continue;
        if (n_pops < 2 && (k == B9_P_Y2 || k == B9_P_LAMBDA)) continue;
        if (pr.var[k] > 0.0) {
            double d = par[k] - pr.mean[k];
            lp -= 0.5 * d * d / pr.var[k];
        }
    }
    return lp;
}

// block-wide sum of one int per thread (all threads get the result); blockDim.x = 256
__device__ __forceinline__ int block_count(bool pred, int *s_cnt)
{
    const int tid = threadIdx.x;
    const int c = __popcll(__ballot(pred));
    __syncthreads();                       // s_cnt may still be read from the previous round
    if ((tid & 63) == 0) s_cnt[tid >> 6] = c;
    __syncthreads();
    return (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
}

// log-posterior of walker w from its partials; all 256 threads call, all get the value.
// The summation order is fixed (thread-strided, wave shuffle tree, four wave totals in order), so
// every workgroup that calls this for the same walker obtains the same bits.
__device__ __forceinline__ double finish_logpost(const IsoHdr *__restrict__ hdr, const double *__restrict__ partial,
                                                 int n_partial, const double *__restrict__ par_row,
                                                 const DevPriors &pr, int n_pops, int w, double *s_red,
                                                 bool *in_support = nullptr)
{
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int j = tid; j < n_partial; j += 256) acc += partial[j];
    bool valid = true;
    for (int k = 0; k < n_pops; ++k) valid = valid && hdr[w * n_pops + k].valid;
    const double lp = log_prior_cluster(pr, par_row, n_pops);
    const double sum = wave_sum(acc);
    __syncthreads();                       // s_red may still be read by an earlier use
    if ((tid & 63) == 0) s_red[tid >> 6] = sum;
    __syncthreads();
    const double t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    if (in_support) *in_support = valid && lp != NEG_INF;      // inside the grid and the prior's support
    return (valid && lp != NEG_INF) ? lp + t : NEG_INF;
}

// Metropolis accept/reject of walker w's proposal `prop_row` with log-posterior lp_prop
// (SURVEY 8f row 1).  All threads call (the decision is needed by all); the new state lands in
// s_cur[12] / *s_lp (LDS); `writer` workgroups also store it to the other state half and append
// the chain record.  u comes from the walker's Philox stream (draw index n_pairs of step `step`).
__device__ __forceinline__ void metropolis_accept(const McmcDev &mc, int w, unsigned long long step, int row,
                                                  const double *__restrict__ prop_row, double lp_prop,
                                                  bool writer, double *s_cur, double *s_lp)
{
    const int tid = threadIdx.x;
    unsigned r[4];
    philox4x32((unsigned)step, (unsigned)(step >> 32), (unsigned)mc.walker_ids[w], (unsigned)((mc.d + 1) >> 1), mc.k0, mc.k1, r);
    const double u = u01(r[0], r[1]);
    const double *cur_in = mc.cur + ((size_t)mc.pin * mc.n_walkers + w) * B9_NPARAM;
    const double lp_cur = mc.lp_cur[(size_t)mc.pin * mc.n_walkers + w];
    const bool ok = isfinite(lp_prop) && (log(u) < lp_prop - lp_cur);
    __syncthreads();
    if (tid < B9_NPARAM) s_cur[tid] = ok ? prop_row[tid] : cur_in[tid];
    if (tid == 0) *s_lp = ok ? lp_prop : lp_cur;
    __syncthreads();
    if (writer) {
        double *cur_out = mc.cur + ((size_t)(mc.pin ^ 1) * mc.n_walkers + w) * B9_NPARAM;
        if (tid < B9_NPARAM) cur_out[tid] = s_cur[tid];
        if (tid == 0) {
            mc.lp_cur[(size_t)(mc.pin ^ 1) * mc.n_walkers + w] = *s_lp;
            if (ok) atomicAdd(mc.n_acc, 1ull);
            if (mc.lps) mc.lps[(size_t)row * mc.n_walkers + w] = *s_lp;
        }
        if (mc.samples && tid < mc.d) mc.samples[((size_t)row * mc.n_walkers + w) * mc.d + tid] = s_cur[mc.free_idx[tid]];
    }
}

// Standard normals of walker w's proposal for step `step` (Philox + Box-Muller) into s_z, by
// threads [t0, t0 + n_pairs).  Depends only on (seed, step, walker): k_derive_iso issues it at
// kernel entry, on a wave that is otherwise idle while the previous step is being finished.
__device__ __forceinline__ void draw_z(const McmcDev &mc, int w, unsigned long long step, int t0, double *s_z)
{
    const int j = (int)threadIdx.x - t0, n_pairs = (mc.d + 1) >> 1;
    if (j >= 0 && j < n_pairs) {
        unsigned r[4];
        philox4x32((unsigned)step, (unsigned)(step >> 32), (unsigned)mc.walker_ids[w], (unsigned)j, mc.k0, mc.k1, r);
        const double u1 = u01(r[0], r[1]), u2 = u01(r[2], r[3]);
        const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
        s_z[2 * j] = rad * cos(ang);
        s_z[2 * j + 1] = rad * sin(ang);
    }
}

// ------------------------------------------------------------------------------------------
// k_derive_iso
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
struct Corners {
    long long off[8];      // point offset of EEP `lo` in each corner isochrone
    int ny;
    double t_age, t_y, t_feh;
};

template <bool MASS>
__device__ __forceinline__ double interp_corner(const DevPack &pk, const Corners &c, int e, int col)
{
    double vf[2];
#pragma unroll
    for (int df = 0; df < 2; ++df) {
        double vy[2] = {0.0, 0.0};
        for (int dy = 0; dy < c.ny; ++dy) {
            long long p0 = c.off[(df * 2 + dy) * 2 + 0] + e, p1 = c.off[(df * 2 + dy) * 2 + 1] + e;
            double a = MASS ? pk.mass[p0] : pk.mags[p0 * pk.nfp + col];
            double b = MASS ? pk.mass[p1] : pk.mags[p1 * pk.nfp + col];
            vy[dy] = lerp(a, b, c.t_age);
        }
        vf[df] = (c.ny == 2) ? lerp(vy[0], vy[1], c.t_y) : vy[0];
    }
    return lerp(vf[0], vf[1], c.t_feh);
}

// Largest i in [0, n-2] with ax[i] <= x, found by one wave in one step: lane l loads ax[l]
// (axes have <= 64 * B9_AXIS_CHUNKS entries) and the bracket is a popcount of the ballot.
// Equal to the oracle's bracket() for an ascending axis.
__device__ __forceinline__ int bracket_wave(const double *__restrict__ ax, int n, double x, int lane)
{
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const bool le = (j < n) && (ax[j] <= x);
        cnt += __popcll(__ballot(le));
    }
    int i = cnt - 1;
    return i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
}

// The three grid axes, one per wave (0: logAge, 1: FeH, 2: Y), preloaded into registers: lane l of
// the wave holds ax[l] and ax[l + 64].  Loading them needs no parameter, so k_derive_iso requests
// them at kernel entry, in the same round trip as everything else it reads first.
struct AxisRegs { double v0, v1; int n; };

__device__ __forceinline__ AxisRegs preload_axis(const DevPack &pk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *ax = wave == 0 ? pk.log_age : (wave == 1 ? pk.feh : pk.y);
    AxisRegs a;
    a.n = wave == 0 ? pk.n_age : (wave == 1 ? pk.n_feh : (wave == 2 ? pk.n_y : 0));
    a.v0 = lane < a.n ? ax[lane] : __builtin_inf();
    a.v1 = lane + 64 < a.n ? ax[lane + 64] : __builtin_inf();
    return a;
}

// bracket of x on a preloaded axis (n <= 128), else on the axis in memory
__device__ __forceinline__ int bracket_regs(const AxisRegs &a, const double *__restrict__ ax, double x, int lane)
{
    if (a.n > 128) return bracket_wave(ax, a.n, x, lane);
    const int cnt = __popcll(__ballot(a.v0 <= x)) + __popcll(__ballot(a.v1 <= x));
    const int i = cnt - 1;
    return i < 0 ? 0 : (i > a.n - 2 ? a.n - 2 : i);
}

// Derives the isochrone of (walker w, population pop) from parameter row `par` (any address
// space).  All threads of the workgroup call it; workgroup `part` of `parts` produces its share of
// the output values (one value per thread and iteration) and part 0 publishes the header.
// Three dependent round trips: {parameters, axes} -> corner index rows -> table values.
__device__ __forceinline__ void derive_iso_block(const DevPack &pk, const double *par, int pop, int wp,
                                                 IsoHdr *__restrict__ hdr, double *__restrict__ iso_data,
                                                 long long iso_stride, int mass_cap, int part, int parts,
                                                 const AxisRegs &axr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
    __shared__ IsoHdr sh;
    __shared__ Corners sc;
    __shared__ int s_br[3];
    const double log_age = par[B9_P_LOGAGE], feh = par[B9_P_FEH];
    const double y = pop ? par[B9_P_Y2] : par[B9_P_Y];
    __syncthreads();                                     // sh / sc / s_br may still be in use (previous call)
    // three waves bracket the three axes concurrently
    if (wave == 0) { int i = bracket_regs(axr, pk.log_age, log_age, lane); if (lane == 0) s_br[0] = i; }
    if (wave == 1) { int i = bracket_regs(axr, pk.feh, feh, lane); if (lane == 0) s_br[1] = i; }
    if (wave == 2) { int i = pk.n_y > 1 ? bracket_regs(axr, pk.y, y, lane) : 0; if (lane == 0) s_br[2] = i; }
    __syncthreads();
    if (wave == 0) {
        // lanes 0..7: one corner isochrone each
        const int ny = pk.n_y > 1 ? 2 : 1;
        const int i_age = s_br[0], i_feh = s_br[1], i_y = s_br[2];
        const int df = (lane >> 2) & 1, dy = (lane >> 1) & 1, da = lane & 1;
        const int dyc = dy < ny ? dy : 0;
        const int kk = ((i_feh + df) * pk.n_y + (i_y + dyc)) * pk.n_age + i_age + da;
        int f0 = -2147483647, f1 = 2147483647;
        long long off = 0;
        double ax0 = 0.0;
        if (lane < 8) { f0 = pk.first[kk]; f1 = f0 + pk.cnt[kk]; off = pk.off[kk]; }
        // lanes 8..13 fetch the axis values the interpolation weights need (same round trip)
        if (lane == 8)  ax0 = pk.log_age[i_age];
        if (lane == 9)  ax0 = pk.log_age[i_age + 1];
        if (lane == 10) ax0 = pk.feh[i_feh];
        if (lane == 11) ax0 = pk.feh[i_feh + 1];
        if (lane == 12) ax0 = pk.y[i_y];
        if (lane == 13) ax0 = pk.y[ny == 2 ? i_y + 1 : i_y];
        int lo = f0, hi = f1;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
            lo = l2 > lo ? l2 : lo;
            hi = h2 < hi ? h2 : hi;
        }
        lo = __shfl(lo, 0, 64); hi = __shfl(hi, 0, 64);
        const double a_lo = __shfl(ax0, 8, 64), a_hi = __shfl(ax0, 9, 64);
        const double f_lo = __shfl(ax0, 10, 64), f_hi = __shfl(ax0, 11, 64);
        const double y_lo = __shfl(ax0, 12, 64), y_hi = __shfl(ax0, 13, 64);
        const double a_min = pk.log_age[0], a_max = pk.log_age[pk.n_age - 1];
        const double f_min = pk.feh[0], f_max = pk.feh[pk.n_feh - 1];
        const double y_min = pk.y[0], y_max = pk.y[pk.n_y - 1];
        if (lane < 8) sc.off[lane] = off + (lo - f0);
        if (lane == 0) {
            IsoHdr h;
            h.valid = 0; h.first_eep = 0; h.n = 0; h.i_feh = i_feh; h.i_y = i_y; h.i_age = i_age;
            h.agb_tip = 0.0; h.t_feh = h.t_y = h.t_age = 0.0;
            bool ok = (log_age >= a_min && log_age <= a_max) && (feh >= f_min && feh <= f_max) &&
                      pk.n_age >= 2 && pk.n_feh >= 2;
            if (pk.n_y > 1) ok = ok && (y >= y_min && y <= y_max);
            const int n = hi - lo;
            if (ok && n >= 2 && n <= mass_cap) {
                h.t_age = (log_age - a_lo) / (a_hi - a_lo);
                h.t_feh = (feh - f_lo) / (f_hi - f_lo);
                if (ny == 2) h.t_y = (y - y_lo) / (y_hi - y_lo);
                h.first_eep = lo; h.n = n; h.valid = 1;
            }
            sc.ny = ny; sc.t_age = h.t_age; sc.t_y = h.t_y; sc.t_feh = h.t_feh;
            sh = h;
        }
    }
    __syncthreads();
    if (!sh.valid) { if (tid == 0 && part == 0) hdr[wp] = sh; return; }
    const int n = sh.n, nfp = pk.nfp;
    double *omass = iso_data + (size_t)wp * iso_stride;
    double *omags = omass + mass_cap;
    // the thread that starts first also interpolates the last point's mass = the AGB-tip mass
    if (part == 0 && tid == 0) {
        IsoHdr h = sh;
        h.agb_tip = interp_corner<true>(pk, sc, n - 1, 0);
        hdr[wp] = h;
    }
    const int total = n * (nfp + 1);
    for (int idx = part * nthreads + tid; idx < total; idx += parts * nthreads) {
        const int e = idx / (nfp + 1), c = idx - e * (nfp + 1);
        if (c == nfp) omass[e] = interp_corner<true>(pk, sc, e, 0);
        else omags[(size_t)e * nfp + c] = (c < pk.nf) ? interp_corner<false>(pk, sc, e, c) : 0.0;
    }
}

// k_derive_iso: grid = (walkers * pops, parts).  Every workgroup of a row re-derives the (cheap)
// header and then produces its share of the values, so the table reads of one isochrone are a
// single round trip spread over ~15 workgroups.
//
// Device-resident sampler (mc.enabled): the launch of step t first finishes step t-1 when
// mc.has_prev -- each workgroup re-sums walker w's partials of the star kernel's previous launch,
// adds the prior of the previous proposal (params_prev) and accepts or rejects it (identical bits
// in every workgroup; workgroup (pop 0, part 0) stores the new state and the chain record) -- and
// then draws step t's proposal from that state, publishes it to `params`, and derives its
// isochrone(s).  One MCMC step = this launch + the star-likelihood launch.
__global__ __launch_bounds__(256) void k_derive_iso(DevPack pk, double *__restrict__ params,
                                                     int n_pops, IsoHdr *__restrict__ hdr,
                                                     double *__restrict__ iso_data, long long iso_stride,
                                                     int mass_cap, McmcDev mc, DevPriors pr,
                                                     const double *__restrict__ partial_prev, int n_partial,
                                                     long long partial_stride,
                                                     const IsoHdr *__restrict__ hdr_prev,
                                                     const double *__restrict__ params_prev)
{
    const int wp = blockIdx.x, w = wp / n_pops, pop = wp % n_pops;
    const double *par = params + (size_t)w * B9_NPARAM;
    __shared__ double s_par[B9_NPARAM], s_z[12], s_cur[B9_NPARAM], s_prop[B9_NPARAM], s_red[4];
    const AxisRegs axr = preload_axis(pk);                 // first round trip, needs no parameter
    if (mc.enabled) {
        // Everything the prologue reads is requested NOW, in one round trip: the walker's current
        // row and log-posterior, the previous proposal, this thread's row of the proposal factor,
        // and (inside finish_logpost) the partials.  Nothing below waits on memory again until the
        // isochrone tables.
        const int tid = threadIdx.x, d = mc.d;
        const bool writer = (blockIdx.y == 0 && pop == 0);
        const size_t st_in = (size_t)mc.pin * mc.n_walkers + w;
        const double cur_v = tid < B9_NPARAM ? mc.cur[st_in * B9_NPARAM + tid] : 0.0;
        const double prop_v = (mc.has_prev && tid < B9_NPARAM) ? params_prev[(size_t)w * B9_NPARAM + tid] : 0.0;
        const double lp_cur = mc.lp_cur[st_in];
        double crow[11];
#pragma unroll
        for (int j = 0; j < 11; ++j) crow[j] = (tid < d && j < d) ? mc.chol[tid * d + j] : 0.0;
        const int fidx = tid < d ? mc.free_idx[tid] : 0;
        draw_z(mc, w, mc.step, 192, s_z);                  // wave 3: this step's normals, independent of the state
        if (tid < B9_NPARAM) { s_prop[tid] = prop_v; s_cur[tid] = cur_v; }
        __syncthreads();
        if (mc.has_prev) {
            const double lp_prop = finish_logpost(hdr_prev, partial_prev + (size_t)w * partial_stride, n_partial,
                                                  s_prop, pr, n_pops, w, s_red);
            // Metropolis accept/reject of step t-1 (u: draw index n_pairs of that step's Philox stream)
            unsigned r[4];
            const unsigned long long sp = mc.step - 1;
            philox4x32((unsigned)sp, (unsigned)(sp >> 32), (unsigned)mc.walker_ids[w], (unsigned)((d + 1) >> 1), mc.k0, mc.k1, r);
            const bool ok = isfinite(lp_prop) && (log(u01(r[0], r[1])) < lp_prop - lp_cur);
            const double lp_new = ok ? lp_prop : lp_cur;
            if (tid < B9_NPARAM && ok) s_cur[tid] = prop_v;      // own slot only: no hazard with the reads above
            __syncthreads();
            if (writer) {
                const size_t st_out = (size_t)(mc.pin ^ 1) * mc.n_walkers + w;
                if (tid < B9_NPARAM) mc.cur[st_out * B9_NPARAM + tid] = s_cur[tid];
                if (tid == 0) {
                    mc.lp_cur[st_out] = lp_new;
                    if (ok) atomicAdd(mc.n_acc, 1ull);
                    if (mc.lps) mc.lps[(size_t)mc.row * mc.n_walkers + w] = lp_new;
                }
                if (mc.samples && tid < d) mc.samples[((size_t)mc.row * mc.n_walkers + w) * d + tid] = s_cur[fidx];
            }
        }
        // proposal of step t:  s_par = state;  s_par[free[i]] += sum_j chol[i][j] z_j  (j ascending, plain multiply-add)
        if (tid < B9_NPARAM) s_par[tid] = s_cur[tid];
        double delta = 0.0;
#pragma unroll
        for (int j = 0; j < 11; ++j) if (j < d) delta = delta + crow[j] * s_z[j];
        __syncthreads();
        if (tid < d) s_par[fidx] += delta;
        __syncthreads();
        if (writer && tid < B9_NPARAM) params[(size_t)w * B9_NPARAM + tid] = s_par[tid];
        par = s_par;
    }
    derive_iso_block(pk, par, pop, wp, hdr, iso_data, iso_stride, mass_cap, blockIdx.y, gridDim.y, axr);
}

// ------------------------------------------------------------------------------------------
// per-star evolution (device functions)
// ------------------------------------------------------------------------------------------
template <int NFP>
struct IsoView {
    const double *mass;   // LDS or global
    const double *mags;   // rows of NFP doubles
    int n;
    double tip;
    int i_feh, i_y;
    double t_feh, t_y;
};

template <int NFP>
__device__ __forceinline__ void fill(double (&out)[NFP], double v)
{
#pragma unroll
    for (int f = 0; f < NFP; ++f) out[f] = v;
}

// binary search in the isochrone's mass column + linear interpolation.
template <int NFP>
__device__ __forceinline__ void msrgb_mags(const IsoView<NFP> &iso, double m, double (&out)[NFP])
{
    if (m < iso.mass[0]) { fill<NFP>(out, B9_MAG_NOFLUX); return; }
    int lo = 0, hi = iso.n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (iso.mass[mid] <= m) lo = mid; else hi = mid;
    }
    const double a = iso.mass[lo], d = iso.mass[lo + 1] - a;
    const double t = (d > 0.0) ? (m - a) / d : 0.0;
    const double *r0 = iso.mags + (size_t)lo * NFP;
#pragma unroll
    for (int f = 0; f < NFP; ++f) out[f] = lerp(r0[f], r0[NFP + f], t);
}

__device__ __forceinline__ double ifmr(int id, const double *__restrict__ par, double m)
{
    switch (id) {
    case B9_IFMR_WEIDEMANN: {
        const double mf[7] = {0.55, 0.60, 0.68, 0.79, 0.88, 0.95, 1.02};
        int i = (int)floor(m) - 1;
        i = i < 0 ? 0 : (i > 5 ? 5 : i);
        // same bracket as the oracle: largest i with mi[i] <= m, clamped to [0, 5]
        double mi = (double)(i + 1);
        return lerp(mf[i], mf[i + 1], (m - mi) / ((double)(i + 2) - mi));
    }
    case B9_IFMR_WILLIAMS:    return 0.339 + 0.129 * m;
    case B9_IFMR_SALARIS_LIN: return 0.466 + 0.084 * m;
    case B9_IFMR_SALARIS_PW:  return (m < 4.0) ? 0.134 * m + 0.331 : 0.047 * m + 0.679;
    case B9_IFMR_LINEAR:      return par[B9_P_IFMR_INTERCEPT] + par[B9_P_IFMR_SLOPE] * (m - 3.0);
    default: {
        double d = m - 3.0;
        return par[B9_P_IFMR_INTERCEPT] + par[B9_P_IFMR_SLOPE] * d + par[B9_P_IFMR_QUAD] * d * d;
    }
    }
}

// Axes the WD branch searches, staged in LDS by k_finalize (a dozen dependent bracket steps per
// star: ~64-cycle ds_reads instead of L2/HBM round trips).  Pointers fall back to global memory
// when the axes do not fit.
struct WdAxes {
    const double *log_age;        // [n_age]
    const double *tips[4];        // [(df*2+dy)][n_age] AGB-tip mass of the corner (FeH, Y) columns
    const double *wc_log_age, *wc_mass, *wc_carb, *at_log_teff, *at_logg;
};

__device__ inline double prec_log_age_corner(const DevPack &pk, const WdAxes &ax, int corner, double m)
{
    const int na = pk.n_age;
    const double *tips = ax.tips[corner];
    const double tip0 = tips[0];
    if (m > tip0) return ax.log_age[0] - 2.7 * log10(m / tip0);
    if (m <= tips[na - 1]) return ax.log_age[na - 1];
    int lo = 0, hi = na - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (tips[mid] >= m) lo = mid; else hi = mid;
    }
    const double a = tips[lo], b = tips[lo + 1];
    const double t = (b != a) ? (m - a) / (b - a) : 0.0;
    return lerp(ax.log_age[lo], ax.log_age[lo + 1], t);
}

// IFMR -> WD cooling model -> atmosphere table.  Only the stars above the AGB
// tip take this branch; k_finalize runs it (the hot kernel never does).
template <int NFP>
__device__ __forceinline__ void wd_mags(const DevPack &pk, const WdAxes &ax, const IsoView<NFP> &iso,
                                     const double *__restrict__ par, double m, int wd_type,
                                     double (&out)[NFP])
{
    if (pk.n_wc_mass < 2 || pk.n_at_teff < 2) { fill<NFP>(out, B9_MAG_NOFLUX); return; }
    const int ny = pk.n_y > 1 ? 2 : 1;
    double vf[2];
    for (int df = 0; df < 2; ++df) {
        double vy[2] = {0.0, 0.0};
        for (int dy = 0; dy < ny; ++dy) vy[dy] = prec_log_age_corner(pk, ax, df * 2 + dy, m);
        vf[df] = (ny == 2) ? lerp(vy[0], vy[1], iso.t_y) : vy[0];
    }
    const double prec = lerp(vf[0], vf[1], iso.t_feh);
    const double log_age = par[B9_P_LOGAGE];
    if (prec >= log_age) { fill<NFP>(out, -4.0); return; }
    const double wd_mass = ifmr(pk.ifmr_id, par, m);
    const double log_cool = log10(exp10(log_age) - exp10(prec));

    const int ia = bracket(ax.wc_log_age, pk.n_wc_age, log_cool);
    const double ta = (log_cool - ax.wc_log_age[ia]) / (ax.wc_log_age[ia + 1] - ax.wc_log_age[ia]);
    const int im = bracket(ax.wc_mass, pk.n_wc_mass, wd_mass);
    const double tm = (wd_mass - ax.wc_mass[im]) / (ax.wc_mass[im + 1] - ax.wc_mass[im]);
    const int nc = pk.n_wc_carb > 1 ? 2 : 1;
    int ic = 0; double tc = 0.0;
    if (nc == 2) {
        ic = bracket(ax.wc_carb, pk.n_wc_carb, par[B9_P_CARBONICITY]);
        tc = (par[B9_P_CARBONICITY] - ax.wc_carb[ic]) / (ax.wc_carb[ic + 1] - ax.wc_carb[ic]);
    }
    double tr[2];
    for (int q = 0; q < 2; ++q) {
        const double *tab = q ? pk.wc_log_radius : pk.wc_log_teff;
        double vc[2] = {0.0, 0.0};
        for (int dc = 0; dc < nc; ++dc) {
            double vm[2];
            for (int dm = 0; dm < 2; ++dm) {
                size_t base = ((size_t)(ic + dc) * pk.n_wc_mass + (im + dm)) * pk.n_wc_age + ia;
                vm[dm] = lerp(tab[base], tab[base + 1], ta);
            }
            vc[dc] = lerp(vm[0], vm[1], tm);
        }
        tr[q] = (nc == 2) ? lerp(vc[0], vc[1], tc) : vc[0];
    }
    const double log_teff = tr[0];
    const double logg = LOG_G_PLUS_LOG_MSUN + log10(wd_mass) - 2.0 * tr[1];
    const int ty = (wd_type > 0 && pk.n_at_type > 1) ? 1 : 0;
    const int it = bracket(ax.at_log_teff, pk.n_at_teff, log_teff);
    const double tt = (log_teff - ax.at_log_teff[it]) / (ax.at_log_teff[it + 1] - ax.at_log_teff[it]);
    const int ig = bracket(ax.at_logg, pk.n_at_logg, logg);
    const double tg = (logg - ax.at_logg[ig]) / (ax.at_logg[ig + 1] - ax.at_logg[ig]);
    const double *g0 = pk.at_mags + (((size_t)ty * pk.n_at_logg + ig) * pk.n_at_teff + it) * NFP;
    const double *g1 = g0 + (size_t)pk.n_at_teff * NFP;
#pragma unroll
    for (int f = 0; f < NFP; ++f) {
        double v0 = lerp(g0[f], g0[NFP + f], tt);
        double v1 = lerp(g1[f], g1[NFP + f], tt);
        out[f] = lerp(v0, v1, tg);
    }
}

// which branch a ZAMS mass is on ([RECALL] Star::getStatus) -- the general form, used by k_finalize
// for the stars the hot kernel skips (hot_star below is the MS/RGB-only form).
template <int NFP>
__device__ __forceinline__ void star_mags(const DevPack &pk, const WdAxes &ax, const IsoView<NFP> &iso,
                                          const double *__restrict__ par, double m, int wd_type,
                                          double (&out)[NFP])
{
    if (!(m > 0.0)) { fill<NFP>(out, B9_MAG_NOFLUX); return; }
    if (m <= iso.tip) { msrgb_mags<NFP>(iso, m, out); return; }
    if (m <= pk.m_wd_up) wd_mags<NFP>(pk, ax, iso, par, m, wd_type, out);
    else fill<NFP>(out, B9_MAG_NOFLUX);
}

// SURVEY 8a rows a5 + a6: combined magnitudes -> sum_f w_f (pred_f - obs_f)^2.
// Flux addition is done as  m1 - 2.5 log10(1 + 10^(-0.4 (m2 - m1)))  : one exp and one log1p
// per filter instead of two pow and a log10, and no cancellation.
template <int NFP>
__device__ __forceinline__ double chi2_system(const DevPack &pk, const WdAxes &ax, const IsoView<NFP> &iso,
                                              const double *__restrict__ par, double m1, double q,
                                              int wd_type, const DevStars &st, int i)
{
    double p1[NFP];
    star_mags<NFP>(pk, ax, iso, par, m1, wd_type, p1);
    if (q > 0.0) {
        double p2[NFP];
        star_mags<NFP>(pk, ax, iso, par, q * m1, wd_type, p2);
#pragma unroll
        for (int f = 0; f < NFP; ++f)
            p1[f] -= (2.5 / LN10) * log1pexp((-0.4 * LN10) * (p2[f] - p1[f]));
    }
    const double mod = par[B9_P_MOD], av = par[B9_P_ABS];
    double chi2 = 0.0;
#pragma unroll
    for (int f = 0; f < NFP; ++f) {
        const double pred = p1[f] + (mod + pk.abs_m1[f] * av);
        const double d = pred - st.obs[(size_t)f * st.n_pad + i];
        chi2 = fma(st.w[(size_t)f * st.n_pad + i] * d, d, chi2);
    }
    // a non-finite predicted magnitude (NaN or inf, also under a zero weight: 0 * inf = NaN)
    // leaves chi2 non-finite: the star is impossible under this isochrone
    return isfinite(chi2) ? chi2 : __builtin_inf();
}

// one star, all populations, field-star mixture: log( (1-p) fsLike + p L_i )
template <int NFP, int NPOPS>
__device__ __forceinline__ double star_value(const DevPack &pk, const WdAxes (&ax)[NPOPS], const IsoView<NFP> (&iso)[NPOPS],
                                             const double *__restrict__ par, const DevStars &st, int i,
                                             double log_lam, double log_1ml)
{
    const double m1 = st.mass1[i], q = st.q[i], c0 = st.c0[i], la = st.la[i];
    const int wd_type = st.flags[i] & 1;
    double ll[NPOPS];
#pragma unroll
    for (int k = 0; k < NPOPS; ++k)
        ll[k] = c0 - 0.5 * chi2_system<NFP>(pk, ax[k], iso[k], par, m1, q, wd_type, st, i);
    double l = ll[0];
    if (NPOPS == 2) l = logaddexp(log_lam + ll[0], log_1ml + ll[NPOPS - 1]);
    return logaddexp(la, l);
}

template <int NFP, int NPOPS>
__device__ __forceinline__ bool load_iso_views(const IsoHdr *__restrict__ hdr, const double *__restrict__ iso_data,
                                               long long iso_stride, int mass_cap, int w,
                                               IsoView<NFP> (&iso)[NPOPS], double &tip_min)
{
    bool valid = true;
    tip_min = __builtin_inf();
#pragma unroll
    for (int k = 0; k < NPOPS; ++k) {
        const IsoHdr h = hdr[w * NPOPS + k];
        valid = valid && h.valid;
        iso[k].n = h.n; iso[k].tip = h.agb_tip;
        iso[k].i_feh = h.i_feh; iso[k].i_y = h.i_y; iso[k].t_feh = h.t_feh; iso[k].t_y = h.t_y;
        const double *g = iso_data + (size_t)(w * NPOPS + k) * iso_stride;
        iso[k].mass = g; iso[k].mags = g + mass_cap;
        tip_min = h.agb_tip < tip_min ? h.agb_tip : tip_min;
    }
    return valid;
}

// ------------------------------------------------------------------------------------------
// k_star_like  (given-mass mode): the hot kernel.  One lane per star, MS/RGB branch only --
// stars heavier than the walker's AGB tip (WD / NS-BH branch; two contiguous ranges because
// stars are sorted by mass) are left to k_finalize so that pow/log10 and the WD tables do not
// cost this kernel registers.
//
// Workgroup -> (star tile, walker) map is XCD-aware: workgroups are dealt round-robin over the 8
// XCDs, so linear id L runs on XCD L % 8.  All walkers of one star tile are given ids with the
// same L % 8 and consecutive L / 8: the tile's star data is fetched from HBM once into that
// XCD's L2 and re-read from L2 by the other walkers.  (Placement affects speed only.)
// ------------------------------------------------------------------------------------------
#ifndef B9_K1_MIN_WAVES
#define B9_K1_MIN_WAVES 3
#endif
#ifndef B9_K1_MIN_WAVES_2POP
#define B9_K1_MIN_WAVES_2POP 2      // two populations: the loop-carried state pushes the body past 168 VGPRs
#endif

// Bracket of mass m in an LDS-resident mass column: the largest i in [0, n-2] with mass[i] <= m
// (what the oracle's binary search returns -- the bracket is unique for a sorted column, so any
// correct search yields the same i and hence bit-identical weights).  8-ary: every step issues 7
// independent ds_reads and narrows the range eightfold, so a 400-point column takes 3 dependent
// LDS round trips instead of the 9 of a binary search (measured: the binary search was 19 % of the
// kernel's VALU instructions but 3.3 of its 20.5 us).
__device__ __forceinline__ void find_bracket(const double *mass, int n, double m, int &lo_out, double &t_out)
{
    int lo = 0, len = n - 1;                 // the answer lies in [lo, lo + len)
    while (len >= 8) {                       // 7 probes at lo + j*step, all inside the range (7*step < len)
        const int step = len >> 3;
        const double *p = mass + lo;
        int c = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) c += (p[j * step] <= m) ? 1 : 0;
        lo += c * step;
        len = (c == 7) ? len - 7 * step : step;
    }
    {                                        // fewer than 8 candidates left: probe them all at once
        const double *p = mass + lo;
        int c = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) c += (j < len && p[j] <= m) ? 1 : 0;     // reads stay inside the column: lo + 7 <= n + 6 < capacity
        lo += c;
    }
    const double a = mass[lo], d = mass[lo + 1] - a;
#ifdef B9_EXACT_DIV
    t_out = (d > 0.0) ? (m - a) / d : 0.0;
#else
    // (m - a) / d by a v_rcp_f64 seed, two Newton steps and a residual correction: within 1 ulp of
    // the IEEE quotient (the weight is then off by <= 1e-16 relative -- seven orders inside the
    // stated tolerance) at a third of the instructions and latency of the exact division sequence
    const double num = m - a;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double tq = num * r;
    tq = fma(fma(-d, tq, num), r, tq);
    t_out = (d > 0.0) ? tq : 0.0;
#endif
    lo_out = lo;
}

#ifndef B9_EARLY_OBS
#define B9_LATE_OBS 1        // measured: 20.3 us vs 22.3 us (early) on the 50k x 8 x 8 bench shape
#endif
#ifdef B9_LATE_OBS
#define B9_OBS_ARGS const DevStars &st, int il, const double *stage
#else
#define B9_OBS_ARGS double c0, const double (&obs)[NFP], const double (&wgt)[NFP]
#endif
// -DB9_QUAD_PASS: filters in passes of four (rows + observations of a pass requested together).
// Measured equal to the all-at-once form (20.7 vs 20.6 us) and only 5 VGPRs leaner -- the pressure
// comes from the unrolled exp/log temporaries, not the row arrays -- so it is off by default.
template <int NFP, int NPOPS>
__device__ __forceinline__ double hot_star(const DevPack &pk, const IsoView<NFP> (&iso)[NPOPS],
                                           double mod, double av, double m1, double q,
                                           B9_OBS_ARGS, double log_lam, double log_1ml)
{
#ifdef B9_ABL_NOBIN
    const bool binary = false;
#else
    const bool binary = q > 0.0;
#endif
    const double m2 = q * m1;
    double ll[2] = {0.0, 0.0};
    // The population loop is deliberately NOT unrolled: unrolled, the compiler overlaps the two
    // populations' row loads and transcendental temporaries and spills (324 B of scratch per lane,
    // 5x slower per star-eval); rolled, the body keeps the single-population register footprint.
    // The isochrone view is picked with wave-uniform selects.
#pragma unroll 1
    for (int k = 0; k < NPOPS; ++k) {
        const double *is_mass = (NPOPS == 2 && k) ? iso[NPOPS - 1].mass : iso[0].mass;
        const double *is_mags = (NPOPS == 2 && k) ? iso[NPOPS - 1].mags : iso[0].mags;
        const int is_n = (NPOPS == 2 && k) ? iso[NPOPS - 1].n : iso[0].n;
        int lo1, lo2 = 0;
        double t1, t2 = 0.0;
        const bool dark1 = !(m1 > 0.0) || m1 < is_mass[0];
        const bool dark2 = !(m2 > 0.0) || m2 < is_mass[0];
#ifdef B9_ABL_NOSEARCH
        lo1 = (int)(m1 * 100.0) % (is_n - 1); t1 = m1 - (int)m1; lo2 = lo1 / 2; t2 = t1;
#else
        find_bracket(is_mass, is_n, m1, lo1, t1);
        if (binary) find_bracket(is_mass, is_n, m2, lo2, t2);
#endif
        STAMP(4);
        // two consecutive rows = 2*NFP contiguous doubles
        const double2 *r1 = reinterpret_cast<const double2 *>(is_mags + (size_t)lo1 * NFP);
        const double2 *r2 = reinterpret_cast<const double2 *>(is_mags + (size_t)lo2 * NFP);
        double chi2 = 0.0;
#if defined(B9_QUAD_PASS) && defined(B9_LATE_OBS)
        // Passes of four filters.  Each pass requests its slice of the primary rows, of the secondary
        // rows and of the observed magnitudes / weights TOGETHER (one round trip per pass, the same
        // two round trips as the all-at-once form at 8 filters), so only a quarter of the row and
        // observation registers are live at a time.
#pragma unroll 1
        for (int h = 0; h < NFP / 4; ++h) {
            double2 a1[4], a2[4];
            a1[0] = r1[2 * h]; a1[1] = r1[2 * h + 1]; a1[2] = r1[NFP / 2 + 2 * h]; a1[3] = r1[NFP / 2 + 2 * h + 1];
            if (binary) { a2[0] = r2[2 * h]; a2[1] = r2[2 * h + 1]; a2[2] = r2[NFP / 2 + 2 * h]; a2[3] = r2[NFP / 2 + 2 * h + 1]; }
            double o[4], wv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = st.obs[(size_t)(4 * h + j) * st.n_pad + il];
                wv[j] = st.w[(size_t)(4 * h + j) * st.n_pad + il];
            }
            double p[4];
            p[0] = dark1 ? B9_MAG_NOFLUX : lerp(a1[0].x, a1[2].x, t1);
            p[1] = dark1 ? B9_MAG_NOFLUX : lerp(a1[0].y, a1[2].y, t1);
            p[2] = dark1 ? B9_MAG_NOFLUX : lerp(a1[1].x, a1[3].x, t1);
            p[3] = dark1 ? B9_MAG_NOFLUX : lerp(a1[1].y, a1[3].y, t1);
            if (binary) {
                double s[4];
                s[0] = dark2 ? B9_MAG_NOFLUX : lerp(a2[0].x, a2[2].x, t2);
                s[1] = dark2 ? B9_MAG_NOFLUX : lerp(a2[0].y, a2[2].y, t2);
                s[2] = dark2 ? B9_MAG_NOFLUX : lerp(a2[1].x, a2[3].x, t2);
                s[3] = dark2 ? B9_MAG_NOFLUX : lerp(a2[1].y, a2[3].y, t2);
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] -= (2.5 / LN10) * log1pexp((-0.4 * LN10) * (s[j] - p[j]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double d = (p[j] + (mod + pk.abs_m1[4 * h + j] * av)) - o[j];
                chi2 = fma(wv[j] * d, d, chi2);
            }
        }
        const double c0 = st.c0[il];
        STAMP(6);
#else
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
#ifdef B9_COMBINE_UNROLL
#pragma unroll B9_COMBINE_UNROLL
#else
#pragma unroll
#endif
            for (int j = 0; j < NFP / 2; ++j) {
                const double s0 = dark2 ? B9_MAG_NOFLUX : lerp(a2[j].x, a2[NFP / 2 + j].x, t2);
                const double s1 = dark2 ? B9_MAG_NOFLUX : lerp(a2[j].y, a2[NFP / 2 + j].y, t2);
                p[2 * j] -= (2.5 / LN10) * log1pexp((-0.4 * LN10) * (s0 - p[2 * j]));
                p[2 * j + 1] -= (2.5 / LN10) * log1pexp((-0.4 * LN10) * (s1 - p[2 * j + 1]));
            }
        }
        STAMP(6);
#ifdef B9_LATE_OBS
        // observed magnitudes and weights are requested only now: they cost 32 VGPRs while live,
        // and keeping them out of the search / row / combine phases buys a wave per SIMD
        __builtin_amdgcn_sched_barrier(0);
        double obs[NFP], wgt[NFP];
        double c0;
        if (stage) {
            // the fused step stages this wave's observed magnitudes, weights and c0 in LDS with asynchronous
            // global->LDS loads issued at the START of the tile (stage_tile): by now they have landed, so
            // this phase costs LDS reads instead of a memory round trip
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int f = 0; f < NFP; ++f) { obs[f] = stage[f * 64]; wgt[f] = stage[(NFP + f) * 64]; }
            c0 = stage[2 * NFP * 64];
        } else {
#pragma unroll
            for (int f = 0; f < NFP; ++f) {
                obs[f] = st.obs[(size_t)f * st.n_pad + il];
                wgt[f] = st.w[(size_t)f * st.n_pad + il];
            }
            c0 = st.c0[il];
        }
#endif
#pragma unroll
        for (int f = 0; f < NFP; ++f) {
            const double d = (p[f] + (mod + pk.abs_m1[f] * av)) - obs[f];
            chi2 = fma(wgt[f] * d, d, chi2);
        }
#endif
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
#ifdef B9_ABL_NOMIX
    const bool additive = true;
#else
    const bool additive = (ea == 0.0) || (l > 600.0);
#endif
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double m2 = __shfl_down(a.mant, o, 64);
        const int e2 = __shfl_down(a.expo, o, 64);
        const double d2 = __shfl_down(a.add, o, 64);
        const double m = a.mant * m2;                      // both in [0.5, 1): product in [0.25, 1)
        a.expo += e2 + __builtin_amdgcn_frexp_exp(m);
        a.mant = __builtin_amdgcn_frexp_mant(m);
        a.add += d2;
    }
    // mant in [0.5, 1): log(mant) = log_ge1(2 mant) - ln 2 (the lean log instead of the library one)
    return (log_ge1(a.mant + a.mant) + (double)(a.expo - 1) * 0.693147180559945309417) + a.add;
}

// The stars the hot path skips -- primary heavier than the walker's AGB tip (SURVEY 8a row a7:
// IFMR -> WD cooling -> WD atmosphere, or NS/BH) -- are evaluated by extra workgroups of the SAME
// launch, through the general per-star code.  Because stars are also indexed by descending mass
// (heavy_mass / heavy_slot), that set is a prefix whose length each heavy workgroup finds with a
// 256-ary search (two rounds for 50k stars).  The WD axes are staged in LDS.  `parts` workgroups
// share a walker's heavy stars; each writes one partial.
template <int NFP, int NPOPS>
__device__ __forceinline__ void heavy_stars(const DevPack &pk, const DevStars &st, const IsoHdr *__restrict__ hdr,
                                         const double *__restrict__ iso_data, long long iso_stride, int mass_cap,
                                         const double *__restrict__ params, int w, int part, int parts,
                                         double *__restrict__ out_partial, double *__restrict__ perstar, double *smem)
{
    const int tid = threadIdx.x;
    int *s_cnt = reinterpret_cast<int *>(smem);         // 4 ints
    double *s_red = smem + 2;                            // 4 doubles
    double *s_axes = smem + 8;
    const double *par = params + (size_t)w * B9_NPARAM;
    IsoView<NFP> iso[NPOPS];
    double tip_min;
    const bool valid = load_iso_views<NFP, NPOPS>(hdr, iso_data, iso_stride, mass_cap, w, iso, tip_min);
    if (!valid) { if (tid == 0) *out_partial = 0.0; return; }
    int lo = 0, hi = st.n;                               // count = first k with heavy_mass[k] <= tip_min
    while (lo < hi) {
        const int span = hi - lo, step = (span + 255) / 256;
        const int p = lo + tid * step;
        const bool above = (p < hi) && (st.heavy_mass[p] > tip_min);
        const int c = block_count(above, s_cnt);
        if (c == 0) { hi = lo; }
        else { const int nlo = lo + (c - 1) * step + 1, nhi = lo + c * step; lo = nlo; hi = nhi < hi ? nhi : hi; }
    }
    const int count = lo;
    double acc = 0.0;
    if (count > 0) {
        WdAxes ax[NPOPS];
        const int na = pk.n_age, ny = pk.n_y > 1 ? 2 : 1;
        double *d = s_axes;
        const double *src[6] = {pk.log_age, pk.wc_log_age, pk.wc_mass, pk.wc_carb, pk.at_log_teff, pk.at_logg};
        const int len[6] = {na, pk.n_wc_age, pk.n_wc_mass, pk.n_wc_carb, pk.n_at_teff, pk.n_at_logg};
        const double *dst[6];
        for (int a = 0; a < 6; ++a) {
            dst[a] = d;
            for (int j = tid; j < len[a]; j += 256) d[j] = src[a][j];
            d += len[a];
        }
        for (int kp = 0; kp < NPOPS; ++kp)               // each population brackets (FeH, Y) on its own
            for (int c = 0; c < 4; ++c) {
                const int df = c >> 1, dy = c & 1;
                const double *tips = pk.tips + (size_t)((iso[kp].i_feh + df) * pk.n_y + (iso[kp].i_y + (dy < ny ? dy : 0))) * na;
                for (int j = tid; j < na; j += 256) d[j] = tips[j];
                ax[kp].tips[c] = d;
                d += na;
            }
        __syncthreads();
        for (int kp = 0; kp < NPOPS; ++kp) {
            ax[kp].log_age = dst[0]; ax[kp].wc_log_age = dst[1]; ax[kp].wc_mass = dst[2]; ax[kp].wc_carb = dst[3];
            ax[kp].at_log_teff = dst[4]; ax[kp].at_logg = dst[5];
        }
        const double lam = NPOPS == 2 ? par[B9_P_LAMBDA] : 1.0;
        const double log_lam = NPOPS == 2 ? log(lam) : 0.0, log_1ml = NPOPS == 2 ? log1p(-lam) : 0.0;
        for (int j = part * 256 + tid; j < count; j += parts * 256) {
            const int i = st.heavy_slot[j];
            const double v = star_value<NFP, NPOPS>(pk, ax, iso, par, st, i, log_lam, log_1ml);
            if (perstar) perstar[(size_t)w * st.n + st.perm[i]] = v;
            acc += v;
        }
    }
    const double sum = wave_sum(acc);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) *out_partial = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

template <int NFP, int NPOPS, int WB>
__global__ __launch_bounds__(256, (NPOPS == 2 && B9_K1_MIN_WAVES > B9_K1_MIN_WAVES_2POP) ? B9_K1_MIN_WAVES_2POP : B9_K1_MIN_WAVES) void k_star_like(DevPack pk, DevStars st,
                                                    const IsoHdr *__restrict__ hdr,
                                                    const double *__restrict__ iso_data,
                                                    long long iso_stride, int mass_cap,
                                                    const double *__restrict__ params, int n_walkers,
                                                    double *__restrict__ partial, long long partial_stride, int n_groups,
                                                    double *__restrict__ perstar, int tiles_per_block,
                                                    int hot_blocks, int heavy_parts)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // Heavy-star workgroups come FIRST in the grid (hot_blocks = their padded count): they have the
    // longest dependent chain, so they must start at once and run beside the hot workgroups.
    if ((int)blockIdx.x < hot_blocks) {        // hot_blocks doubles as "first hot workgroup id"
        const int hb = blockIdx.x;
        if (hb >= n_walkers * heavy_parts) return;            // padding to a multiple of 8
        const int w = hb / heavy_parts, part = hb - w * heavy_parts;
#ifndef B9_ABL_NO_HEAVY
        heavy_stars<NFP, NPOPS>(pk, st, hdr, iso_data, iso_stride, mass_cap, params, w, part, heavy_parts,
                                partial + (size_t)w * partial_stride + (size_t)n_groups * 4 + part, perstar, smem);
#else
        if (threadIdx.x == 0) partial[(size_t)w * partial_stride + (size_t)n_groups * 4 + part] = 0.0;   // ablation build
#endif
        return;
    }
    // LDS: the mass column of each (walker, population) isochrone this workgroup evaluates -- the binary search runs in LDS (dependent ds_reads
    // instead of dependent L2 round trips); the magnitude rows a star needs are then read from L2
    // (coalesced: stars are sorted by mass, so neighbouring lanes hit the same or adjacent rows).
    // A lane keeps its star in registers and evaluates it for WB walkers in turn, so the star data
    // crosses the L2 -> CU fabric once per WB walkers.
    const int tid = threadIdx.x;
    STAMP(0);
    const int n_wsets = (n_walkers + WB - 1) / WB;
    const int L = blockIdx.x - hot_blocks, xcd = L & 7, s = L >> 3;      // hot_blocks is a multiple of 8
    const int wset = s % n_wsets, w0 = wset * WB;
    const int group = (s / n_wsets) * 8 + xcd;          // tile group = tiles_per_block consecutive tiles
    if (group >= n_groups) return;
    const int nwb = (n_walkers - w0) < WB ? (n_walkers - w0) : WB;   // walkers in this set
    const int tile0 = group * tiles_per_block;

    // ---- first round trip: everything that depends only on the kernel arguments ------------
    int i = tile0 * 256 + tid;
    int il = i < st.n_pad ? i : st.n_pad - 1;               // stay inside the padded arrays
    double m1, q, ea;
#ifndef B9_LATE_OBS
    double obs[NFP], wgt[NFP], c0;
#pragma unroll
    for (int f = 0; f < NFP; ++f) {
        obs[f] = st.obs[(size_t)f * st.n_pad + il];
        wgt[f] = st.w[(size_t)f * st.n_pad + il];
    }
    c0 = st.c0[il];
#endif
    m1 = st.mass1[il]; q = st.q[il]; ea = st.ea[il];
    // the mass columns: the source address needs no header field, and copying the full capacity
    // instead of hdr.n entries costs nothing (the tail is never searched)
    double *const lds_mass = smem;
    for (int c = 0; c < nwb * NPOPS; ++c) {
        const double2 *sm = reinterpret_cast<const double2 *>(iso_data + (size_t)(w0 * NPOPS + c) * iso_stride);
        double2 *dm = reinterpret_cast<double2 *>(lds_mass + (size_t)c * mass_cap);
        for (int j = tid; j < mass_cap / 2; j += 256) dm[j] = sm[j];
    }
    // headers and the few parameters the star loop needs (scalar loads, same round trip)
    IsoView<NFP> iso[WB][NPOPS];
    bool valid[WB];
    double tip_min[WB], mod[WB], av[WB], log_lam[WB], log_1ml[WB];
#pragma unroll
    for (int b = 0; b < WB; ++b) {
        const int w = (b < nwb) ? w0 + b : w0;
        const double *par = params + (size_t)w * B9_NPARAM;
        bool ok = b < nwb;
        double tmin = __builtin_inf();
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) {
            const IsoHdr h = hdr[w * NPOPS + kp];
            ok = ok && h.valid;
            iso[b][kp].n = h.n; iso[b][kp].tip = h.agb_tip;
            iso[b][kp].i_feh = h.i_feh; iso[b][kp].i_y = h.i_y; iso[b][kp].t_feh = h.t_feh; iso[b][kp].t_y = h.t_y;
            iso[b][kp].mass = lds_mass + (size_t)(b * NPOPS + kp) * mass_cap;
            iso[b][kp].mags = iso_data + (size_t)(w * NPOPS + kp) * iso_stride + mass_cap;
            tmin = h.agb_tip < tmin ? h.agb_tip : tmin;
        }
        valid[b] = ok; tip_min[b] = tmin;
        mod[b] = par[B9_P_MOD]; av[b] = par[B9_P_ABS];
        const double lam = NPOPS == 2 ? par[B9_P_LAMBDA] : 1.0;
        log_lam[b] = NPOPS == 2 ? log(lam) : 0.0;
        log_1ml[b] = NPOPS == 2 ? log1p(-lam) : 0.0;
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    MixAcc acc[WB];
#pragma unroll
    for (int b = 0; b < WB; ++b) { acc[b].mant = 0.5; acc[b].expo = 1; acc[b].add = 0.0; }   // = 1.0
    for (int t = 0; t < tiles_per_block; ++t) {
        if ((tile0 + t) * 256 >= st.n_pad) break;
        if (t > 0) {
            i = (tile0 + t) * 256 + tid;
            il = i < st.n_pad ? i : st.n_pad - 1;
#ifndef B9_LATE_OBS
#pragma unroll
            for (int f = 0; f < NFP; ++f) {
                obs[f] = st.obs[(size_t)f * st.n_pad + il];
                wgt[f] = st.w[(size_t)f * st.n_pad + il];
            }
            c0 = st.c0[il];
#endif
            m1 = st.mass1[il]; q = st.q[il]; ea = st.ea[il];
        }
        STAMP(3);
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            if (b >= nwb) continue;
            const int w = w0 + b;
            if (!valid[b]) {   // outside the grid: the walker's log-posterior is -inf (k_finalize)
                if (perstar && i < st.n_pad && st.perm[i] >= 0) perstar[(size_t)w * st.n + st.perm[i]] = NEG_INF;
                continue;
            }
            if (i < st.n_pad && !(m1 > tip_min[b])) {     // empty slots hold m1 = +inf
#ifdef B9_LATE_OBS
                const double l = hot_star<NFP, NPOPS>(pk, iso[b], mod[b], av[b], m1, q, st, il, nullptr, log_lam[b], log_1ml[b]);
#else
                const double l = hot_star<NFP, NPOPS>(pk, iso[b], mod[b], av[b], m1, q, c0, obs, wgt, log_lam[b], log_1ml[b]);
#endif
                mix_add(acc[b], ea, l);
                if (perstar) perstar[(size_t)w * st.n + st.perm[i]] = mix_value(ea, l);
            }
        }
    }
    STAMP(7);
    // wave combine (one log per wave and walker); every wave stores its own partial -- no
    // end-of-kernel barrier, so a cheap (single-star) wave never waits for an expensive one
#pragma unroll
    for (int b = 0; b < WB; ++b) {
        const double tot = mix_wave_total(acc[b]);
        if ((tid & 63) == 0 && b < nwb)
            partial[(size_t)(w0 + b) * partial_stride + group * 4 + (tid >> 6)] = valid[b] ? tot : 0.0;
    }
    STAMP(8);
}

// ------------------------------------------------------------------------------------------
// k_mcmc_step: the fused sampler step (see StepDev in b9_device.h).  ONE launch per MCMC step.
//
// Roles by workgroup id:  [heavy-star workgroups][candidate-derivation workgroups][pad to 8][hot].
// Every role starts with step_decide(): the accept/reject decision of the PREVIOUS step, taken
// redundantly by every workgroup of a walker from the same fixed-order sum (identical bits).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ const double *step_state_in(const StepDev &sd, int w)
{
    return sd.state + ((size_t)(sd.set ^ 1) * sd.n_walkers + w) * B9_STATE_STRIDE;
}

// Decision of step t-1 for walker w.  Every WAVE takes it on its own -- lane l adds the partials
// l, l + 64, ... in order, then the shuffle tree -- so there is no LDS traffic and no barrier, all
// waves of all workgroups obtain the same bits, and a wave may use the shortcut below whatever its
// neighbours do.  lp_new = log-posterior of the state after that step (not set on the shortcut).
// SHORTCUT: the caller only needs the 0/1 outcome -- if the walker's writer workgroup (which leads
// the grid) has already published it for this step, take it from there and skip the sum.  Waves
// that start before the writer is done compute it themselves: same bits either way, nobody waits.
#ifndef B9_SHORTCUT
#define B9_SHORTCUT true
#endif
template <bool SHORTCUT>
__device__ __forceinline__ bool step_decide(const StepDev &sd, int w, double &lp_new)
{
    const int lane = threadIdx.x & 63;
    const double *in = step_state_in(sd, w);
    const double lp_cur = in[B9_ST_LP];
    if (!sd.has_prev) { lp_new = lp_cur; return false; }
    if (SHORTCUT) {
        const unsigned long long f = __hip_atomic_load(sd.decided + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((f >> 1) == sd.step) { lp_new = 0.0; return (f & 1ull) != 0; }      // wave-uniform: one word per walker
    }
    const double *part = sd.partial + (size_t)w * sd.partial_stride + (size_t)(sd.set ^ 1) * (sd.partial_stride / 2);
    double acc = 0.0;
    for (int j = lane; j < sd.n_partial; j += 64) acc += part[j];
    const double lpr = in[B9_ST_LPRIOR], lu = in[B9_ST_LOGU];
    STAMP(9);
    const double t = __shfl(wave_sum(acc), 0, 64);
    STAMP(10);
    STAMP(11);
    const double lp_prop = (lpr != NEG_INF) ? lpr + t : NEG_INF;       // prior + sum, as k_finalize forms it
    const bool ok = isfinite(lp_prop) && (lu < lp_prop - lp_cur);
    lp_new = ok ? lp_prop : lp_cur;
    return ok;
}

// Asynchronous global -> LDS staging of one wave's 64 stars of a tile: observed magnitudes, weights, c0
// = 2 NFP + 1 arrays of 64 doubles.  Lanes 0..31 each move 16 bytes per array (global_load_lds_dwordx4:
// the hardware places lane l's data at dst + 16 l, so the 512 bytes land in star order); no VGPR holds
// the data and nothing waits here -- hot_star waits (vmcnt) when it needs them, a few thousand cycles later.
template <int NFP>
__device__ __forceinline__ void stage_tile(const DevStars &st, int slot0 /* first slot of this wave's 64 */, double *dst)
{
    const int lane = threadIdx.x & 63;
    typedef const __attribute__((address_space(1))) void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
    if (lane < 32) {
#pragma unroll
        for (int f = 0; f < NFP; ++f) {
            __builtin_amdgcn_global_load_lds((gptr)(st.obs + (size_t)f * st.n_pad + slot0 + 2 * lane), (lptr)(dst + f * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(st.w + (size_t)f * st.n_pad + slot0 + 2 * lane), (lptr)(dst + (NFP + f) * 64), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((gptr)(st.c0 + slot0 + 2 * lane), (lptr)(dst + 2 * NFP * 64), 16, 0, 0);
    }
}

// Measured on the bench shape: 20.4 us/step with the stage vs 19.6 without (the other waves of the SIMD already
// hide that round trip; the stage adds 17 DMA instructions and two waits per tile).  Kept as a tested
// compile-time option (-DB9_USE_LDS_STAGE), off.
#ifdef B9_USE_LDS_STAGE
#define B9_LDS_STAGE(NFP) ((NFP) <= 8)        // 16 padded filters would need 68 KB per workgroup: not worth the occupancy
#else
#define B9_LDS_STAGE(NFP) false
#endif

// Hot role: k_star_like's body for one walker, with the mass columns and headers of BOTH candidates
// requested before the decision is known (same round trip as the partial sums the decision needs).
template <int NFP, int NPOPS>
__device__ __forceinline__ void step_hot(const DevPack &pk, const DevStars &st, const StepDev &sd, int L,
                                         int tiles_per_block, int n_groups, double *smem)
{
    const int tid = threadIdx.x, W = sd.n_walkers, mass_cap = sd.mass_cap;
    const int xcd = L & 7, s = L >> 3;
    const int w = s % W;
    const int group = (s / W) * 8 + xcd;                 // tile group = tiles_per_block consecutive tiles
    if (group >= n_groups) return;
    // A workgroup's tiles are STRIDED over the slot order (group, group + n_groups, ...): binaries lead that
    // order, so consecutive tiles would give some workgroups only expensive (binary) tiles and others only
    // cheap ones; strided, every workgroup gets its share of both and they finish together.
    // (Only when the launch is a single occupancy round; with several rounds the slots never idle and
    //  contiguous tiles are faster -- measured 84 vs 94 us at 64 walkers.)
    const bool strided = tiles_per_block < 0;
    if (strided) tiles_per_block = -tiles_per_block;
    const int tile0 = strided ? group : group * tiles_per_block, tile_step = strided ? n_groups : 1;
    STAMP(0);
    int i = tile0 * 256 + tid;
    int il = i < st.n_pad ? i : st.n_pad - 1;
    double m1 = st.mass1[il], q = st.q[il], ea = st.ea[il];
    // this wave's LDS stage for the tile's observations (behind the mass columns)
    double *const stage_w = smem + (size_t)(2 * NPOPS) * mass_cap + 8 + (size_t)(tid >> 6) * ((2 * NFP + 1) * 64);
    if (B9_LDS_STAGE(NFP) && tile0 * 256 < st.n_pad) stage_tile<NFP>(st, tile0 * 256 + (tid & ~63), stage_w);
    const size_t rows = (size_t)W * NPOPS;
    const size_t cb0 = (size_t)(sd.set * 2) * rows + (size_t)w * NPOPS;      // candidate 0; candidate 1 is `rows` further
    double *const lds_mass = smem;
    // The 2 * NPOPS mass columns are contiguous in LDS, so flat element f of the copy lands at lds2[f].
    // The first FR * 256 elements travel through registers: their loads are issued HERE, before the
    // decision's partial sums are requested, and written to LDS after it -- one memory round trip
    // for everything instead of two.
    constexpr int FR = 4;
    const int half = mass_cap / 2, total2 = 2 * NPOPS * half;
    double2 *const lds2 = reinterpret_cast<double2 *>(lds_mass);
    auto src2 = [&](int f) -> const double2 * {
        const int c = f / half, j = f - c * half;
        return reinterpret_cast<const double2 *>(sd.cand_iso + (cb0 + (size_t)(c / NPOPS) * rows + (c % NPOPS)) * sd.iso_stride) + j;
    };
    double2 fr[FR];
#pragma unroll
    for (int k = 0; k < FR; ++k) {
        const int f = tid + k * 256;
        fr[k] = f < total2 ? *src2(f) : double2{0.0, 0.0};
    }
    IsoHdr h[2][NPOPS];
    double pmod[2], pav[2], plam[2];
#pragma unroll
    for (int cand = 0; cand < 2; ++cand) {
#pragma unroll
        for (int kp = 0; kp < NPOPS; ++kp) h[cand][kp] = sd.cand_hdr[cb0 + (size_t)cand * rows + kp];
        const double *par = sd.cand_par + ((size_t)(sd.set * 2 + cand) * W + w) * B9_NPARAM;
        pmod[cand] = par[B9_P_MOD]; pav[cand] = par[B9_P_ABS]; plam[cand] = NPOPS == 2 ? par[B9_P_LAMBDA] : 1.0;
    }
    double lp_new;
    STAMP(1);
    const int sel = step_decide<B9_SHORTCUT>(sd, w, lp_new) ? 1 : 0;
    STAMP(2);
#pragma unroll
    for (int k = 0; k < FR; ++k) {
        const int f = tid + k * 256;
        if (f < total2) lds2[f] = fr[k];
    }
    for (int f = tid + FR * 256; f < total2; f += 256) lds2[f] = *src2(f);     // very long isochrones only
    __syncthreads();                                     // the LDS mass columns
    STAMP(3);
    IsoView<NFP> iso[NPOPS];
    bool valid = true;
    double tip_min = __builtin_inf();
#pragma unroll
    for (int kp = 0; kp < NPOPS; ++kp) {
        const IsoHdr hh = sel ? h[1][kp] : h[0][kp];
        valid = valid && hh.valid;
        iso[kp].n = hh.n; iso[kp].tip = hh.agb_tip;
        iso[kp].i_feh = hh.i_feh; iso[kp].i_y = hh.i_y; iso[kp].t_feh = hh.t_feh; iso[kp].t_y = hh.t_y;
        iso[kp].mass = lds_mass + (size_t)(sel * NPOPS + kp) * mass_cap;
        iso[kp].mags = sd.cand_iso + (cb0 + (size_t)sel * rows + kp) * sd.iso_stride + mass_cap;
        tip_min = hh.agb_tip < tip_min ? hh.agb_tip : tip_min;
    }
    const double mod = sel ? pmod[1] : pmod[0], av = sel ? pav[1] : pav[0], lam = sel ? plam[1] : plam[0];
    const double log_lam = NPOPS == 2 ? log(lam) : 0.0, log_1ml = NPOPS == 2 ? log1p(-lam) : 0.0;

    MixAcc acc;
    acc.mant = 0.5; acc.expo = 1; acc.add = 0.0;         // = 1.0
    for (int t = 0; t < tiles_per_block; ++t) {
        if ((tile0 + t * tile_step) * 256 >= st.n_pad) break;
#ifndef B9_NO_TILE_PREFETCH
        // the NEXT tile's star scalars are requested before this tile's arithmetic: one memory round
        // trip less on every tile after the first, for 6 VGPRs
        const int i_n = (tile0 + (t + 1) * tile_step) * 256 + tid;
        const int il_n = i_n < st.n_pad ? i_n : st.n_pad - 1;
        const double m1_n = st.mass1[il_n], q_n = st.q[il_n], ea_n = st.ea[il_n];
#else
        if (t > 0) {
            i = (tile0 + t * tile_step) * 256 + tid;
            il = i < st.n_pad ? i : st.n_pad - 1;
            m1 = st.mass1[il]; q = st.q[il]; ea = st.ea[il];
        }
#endif
        if (valid && i < st.n_pad && !(m1 > tip_min)) {   // empty slots hold m1 = +inf
            const double l = hot_star<NFP, NPOPS>(pk, iso, mod, av, m1, q, st, il,
                                                  B9_LDS_STAGE(NFP) ? stage_w + (tid & 63) : nullptr, log_lam, log_1ml);
            mix_add(acc, ea, l);
        }
#ifndef B9_NO_TILE_PREFETCH
        i = i_n; il = il_n; m1 = m1_n; q = q_n; ea = ea_n;
#endif
        if (B9_LDS_STAGE(NFP) && t + 1 < tiles_per_block && (tile0 + (t + 1) * tile_step) * 256 < st.n_pad) {
            // the next tile's observations: every lane of this wave has consumed the current ones (their
            // ds_reads have returned -- the chi^2 used them), so the stage can be overwritten
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stage_tile<NFP>(st, (tile0 + (t + 1) * tile_step) * 256 + (tid & ~63), stage_w);
        }
    }
    STAMP(7);
    const double tot = mix_wave_total(acc);
    if ((tid & 63) == 0)
        sd.partial[(size_t)w * sd.partial_stride + (size_t)sd.set * (sd.partial_stride / 2) + group * 4 + (tid >> 6)] = valid ? tot : 0.0;
    STAMP(8);
}

// Candidate-derivation role (and, for candidate 0 / population 0 / part 0 of each walker, the
// WRITER of the new state and of the chain record).
__device__ __forceinline__ void step_derive(const DevPack &pk, const StepDev &sd, const DevPriors &pr,
                                            int w, int cand, int pop, int part, int parts)
{
    const int tid = threadIdx.x, d = sd.d, W = sd.n_walkers, n_pops = sd.n_pops;
    __shared__ double s_par[B9_NPARAM], s_z[12], s_cur[B9_NPARAM], s_prop[B9_NPARAM];
    const AxisRegs axr = preload_axis(pk);                 // first round trip, needs no parameter
    // everything the role reads before the isochrone tables is requested now, in one round trip
    const double *in = step_state_in(sd, w);
    const double cur_v = tid < B9_NPARAM ? in[B9_ST_CUR + tid] : 0.0;
    const double prev_prop_v = tid < B9_NPARAM ? in[B9_ST_PROP + tid] : 0.0;
    const size_t rows = (size_t)W * n_pops;
    const double pc0 = tid < B9_NPARAM ? sd.cand_par[((size_t)(sd.set * 2 + 0) * W + w) * B9_NPARAM + tid] : 0.0;
    const double pc1 = tid < B9_NPARAM ? sd.cand_par[((size_t)(sd.set * 2 + 1) * W + w) * B9_NPARAM + tid] : 0.0;
    bool v0 = true, v1 = true;
    for (int k = 0; k < n_pops; ++k) {
        v0 = v0 && sd.cand_hdr[(size_t)(sd.set * 2 + 0) * rows + (size_t)w * n_pops + k].valid;
        v1 = v1 && sd.cand_hdr[(size_t)(sd.set * 2 + 1) * rows + (size_t)w * n_pops + k].valid;
    }
    double crow[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) crow[j] = (tid < d && j < d) ? sd.chol[tid * d + j] : 0.0;
    const int fidx = tid < d ? sd.free_idx[tid] : 0;
    {   // wave 3: the normals of step t+1 (Philox + Box-Muller), independent of every decision
        const int j = tid - 192, n_pairs = (d + 1) >> 1;
        if (j >= 0 && j < n_pairs) {
            unsigned r[4];
            const unsigned long long sn = sd.step + 1;
            philox4x32((unsigned)sn, (unsigned)(sn >> 32), (unsigned)sd.walker_ids[w], (unsigned)j, sd.k0, sd.k1, r);
            const double u1 = u01(r[0], r[1]), u2 = u01(r[2], r[3]);
            const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
            s_z[2 * j] = rad * cos(ang);
            s_z[2 * j + 1] = rad * sin(ang);
        }
    }
    double lp_new;
    const bool ok = step_decide<false>(sd, w, lp_new);
    if (tid < B9_NPARAM) {
        s_cur[tid] = ok ? prev_prop_v : cur_v;             // state after step t-1
        s_prop[tid] = ok ? pc1 : pc0;                      // the proposal THIS launch's star workgroups evaluate
    }
    const bool writer = (cand == 0 && pop == 0 && part == 0);
    if (writer && tid == 0 && sd.has_prev)                 // publish the outcome for workgroups that start later
        __hip_atomic_store(sd.decided + w, (sd.step << 1) | (ok ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (writer) {
        double *out = sd.state + ((size_t)sd.set * W + w) * B9_STATE_STRIDE;
        if (tid < B9_NPARAM) { out[B9_ST_CUR + tid] = s_cur[tid]; out[B9_ST_PROP + tid] = s_prop[tid]; }
        if (tid == 0) {
            out[B9_ST_LP] = lp_new;
            const bool pv = ok ? v1 : v0;
            out[B9_ST_LPRIOR] = pv ? log_prior_cluster(pr, s_prop, n_pops) : NEG_INF;
            out[B9_ST_SEL] = ok ? 1.0 : 0.0;
            {   // log u of the accept test of the proposal evaluated by THIS launch (draw index n_pairs of its step)
                unsigned r[4];
                philox4x32((unsigned)sd.step, (unsigned)(sd.step >> 32), (unsigned)sd.walker_ids[w], (unsigned)((d + 1) >> 1), sd.k0, sd.k1, r);
                out[B9_ST_LOGU] = log(u01(r[0], r[1]));
            }
            if (sd.has_prev) {
                if (ok) atomicAdd(sd.n_acc, 1ull);
                if (sd.lps) sd.lps[(size_t)sd.row * W + w] = lp_new;
            }
        }
        if (sd.has_prev && sd.samples && tid < d) sd.samples[((size_t)sd.row * W + w) * d + tid] = s_cur[fidx];
    }
    if (!sd.derive_next) return;
    // candidate `cand` of step t+1:  base = state (step t rejected) or step t's proposal (accepted);
    // row[free[i]] += sum_j chol[i][j] z_j  (j ascending, plain multiply-add -- as the host twin does)
    if (tid < B9_NPARAM) s_par[tid] = cand ? s_prop[tid] : s_cur[tid];
    double delta = 0.0;
#pragma unroll
    for (int j = 0; j < 11; ++j) if (j < d) delta = delta + crow[j] * s_z[j];
    __syncthreads();
    if (tid < d) s_par[fidx] += delta;
    __syncthreads();
    const size_t cset = (size_t)((sd.set ^ 1) * 2 + cand);
    if (pop == 0 && part == 0 && tid < B9_NPARAM) sd.cand_par[(cset * W + w) * B9_NPARAM + tid] = s_par[tid];
    derive_iso_block(pk, s_par, pop, w * n_pops + pop, sd.cand_hdr + cset * rows, sd.cand_iso + cset * rows * sd.iso_stride,
                     sd.iso_stride, sd.mass_cap, part, parts, axr);
}

// Grid: [heavy-star workgroups][one WRITER per walker][pad to 8][hot workgroups][derivation workgroups].
// The writers lead so that the new state and the published decision exist early; the other
// derivation workgroups trail the grid -- nobody in this launch waits for their output, so they
// fill the slots the last hot workgroups leave free.  (B9_DERIVE_FIRST=1 puts them in front.)
template <int NFP, int NPOPS>
__global__ __launch_bounds__(256, (NPOPS == 2 && B9_K1_MIN_WAVES > B9_K1_MIN_WAVES_2POP) ? B9_K1_MIN_WAVES_2POP : B9_K1_MIN_WAVES)
void k_mcmc_step(DevPack pk, DevStars st, StepDev sd, DevPriors pr, int tiles_per_block, int n_groups,
                 int front_blocks, int hot_blocks, int heavy_parts, int derive_parts, int derive_first)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int W = sd.n_walkers, n_heavy = W * heavy_parts, n_derive = W * 2 * NPOPS * derive_parts;
    int b = blockIdx.x;
    // role of this workgroup: 0 hot, 1 heavy, 2 derivation (index b within the role), 3 none (padding)
    int role;
    if (b < front_blocks) {
        if (b < n_heavy) role = 1;
        else {
            b -= n_heavy;
            if (derive_first) role = b < n_derive ? 2 : 3;
            else if (b < W) { role = 2; b *= 2 * NPOPS * derive_parts; }       // the writer of walker b: (cand 0, pop 0, part 0)
            else role = 3;
        }
    } else {
        b -= front_blocks;
        if (b < hot_blocks) role = 0;
        else {                                                                  // trailing derivation workgroups (derive_first == 0)
            b -= hot_blocks;
            const bool writer = (b % (2 * NPOPS * derive_parts)) == 0;          // those already ran in front
            role = (b < n_derive && !writer && sd.derive_next) ? 2 : 3;
        }
    }
    if (role == 0) { step_hot<NFP, NPOPS>(pk, st, sd, b, tiles_per_block, n_groups, smem); return; }
    if (role == 1) {
        const int w = b / heavy_parts, part = b - w * heavy_parts;
        double lp_new;
        const int sel = step_decide<B9_SHORTCUT>(sd, w, lp_new) ? 1 : 0;
        const size_t rows = (size_t)W * NPOPS, cs = (size_t)(sd.set * 2 + sel);
        heavy_stars<NFP, NPOPS>(pk, st, sd.cand_hdr + cs * rows, sd.cand_iso + cs * rows * sd.iso_stride, sd.iso_stride,
                                sd.mass_cap, sd.cand_par + cs * W * B9_NPARAM, w, part, heavy_parts,
                                sd.partial + (size_t)w * sd.partial_stride + (size_t)sd.set * (sd.partial_stride / 2) +
                                    (size_t)n_groups * 4 + part,
                                nullptr, smem);
        return;
    }
    if (role == 2) {       // b = ((w * 2 + cand) * NPOPS + pop) * derive_parts + part
        const int part = b % derive_parts; b /= derive_parts;
        const int pop = b % NPOPS; b /= NPOPS;
        step_derive(pk, sd, pr, b >> 1, b & 1, pop, part, derive_parts);
    }
}

// the block's last decision: one workgroup per walker, writer role only
__global__ __launch_bounds__(256) void k_mcmc_finish(DevPack pk, StepDev sd, DevPriors pr)
{
    step_derive(pk, sd, pr, blockIdx.x, 0, 0, 0, 1);
}

#ifdef B9_STAMPS
extern "C" int b9_debug_read_stamps(unsigned long long *out, int n_waves)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * B9_NSTAMP * n_waves);
}
extern "C" int b9_debug_clear_stamps(void)
{
    static unsigned long long zeros[8192 * B9_NSTAMP];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof zeros);
}
#endif

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

template <int NFP, int NPOPS, bool SAMPLE>
__global__ __launch_bounds__(256) void k_star_marg(DevPack pk, DevStars st, const IsoHdr *__restrict__ hdr,
                                                    const double *__restrict__ iso_data, long long iso_stride,
                                                    int mass_cap, const double *__restrict__ params,
                                                    double *__restrict__ vals, double *__restrict__ perstar,
                                                    int K, int Q, MargSample ms, int chunk_cap)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.y;
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
            ax.wc_log_age = pk.wc_log_age; ax.wc_mass = pk.wc_mass; ax.wc_carb = pk.wc_carb;
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

// ------------------------------------------------------------------------------------------
// k_finalize: one workgroup per walker: fixed-order sum of the partials + prior -> logpost[w]
// (SURVEY 8a row a8); -inf for a walker outside the grid; with mc.enabled also the accept/reject
// of the block's last step.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(const IsoHdr *__restrict__ hdr, const double *__restrict__ partial,
                                                   int n_partial, long long partial_stride, int n_pops,
                                                   const double *__restrict__ params, DevPriors pr,
                                                   double *__restrict__ logpost, double *__restrict__ perstar,
                                                   int n_stars, McmcDev mc)
{
    __shared__ double s_red[4], s_cur[B9_NPARAM], s_lp;
    const int w = blockIdx.x, tid = threadIdx.x;
    const double *row = params + (size_t)w * B9_NPARAM;
    bool in_support;
    const double lp = finish_logpost(hdr, partial + (size_t)w * partial_stride, n_partial, row, pr, n_pops, w, s_red, &in_support);
    if (tid == 0) logpost[w] = lp;
    // a walker inside the grid whose prior is -inf still has per-star values from the star kernel;
    // the oracle reports -inf for them as well
    if (perstar && !in_support)
        for (int i = tid; i < n_stars; i += 256) perstar[(size_t)w * n_stars + i] = NEG_INF;
    if (mc.enabled) metropolis_accept(mc, w, mc.step, mc.row, row, lp, true, s_cur, &s_lp);
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
hipError_t b9k_derive_iso(const DevPack &pk, double *d_params, int n_walkers, int n_pops,
                          IsoHdr *hdr, double *iso_data, long long iso_stride, int mass_cap,
                          const McmcDev &mc, const DevPriors &pr, const B9Prev &prev, hipStream_t stream)
{
    const int gy = (mass_cap * (pk.nfp + 1) + 255) / 256;
    hipLaunchKernelGGL(k_derive_iso, dim3(n_walkers * n_pops, gy), dim3(256), 0, stream,
                       pk, d_params, n_pops, hdr, iso_data, iso_stride, mass_cap, mc, pr,
                       prev.partial, prev.n_partial, prev.partial_stride, prev.hdr, prev.params);
    return hipGetLastError();
}

static size_t heavy_lds_doubles(const DevPack &pk, int n_pops)
{
    const bool has_wd = pk.n_wc_mass >= 2 && pk.n_at_teff >= 2;
    return 8 + (has_wd ? (size_t)(1 + 4 * n_pops) * pk.n_age + pk.n_wc_age + pk.n_wc_mass + pk.n_wc_carb + pk.n_at_teff + pk.n_at_logg
                       : (size_t)(1 + 4 * n_pops) * pk.n_age);
}

template <int NFP, int NPOPS, int WB>
static hipError_t launch_star_like(const DevPack &pk, const DevStars &st, const IsoHdr *hdr,
                                   const double *iso_data, long long iso_stride, int mass_cap,
                                   const double *d_params, int n_walkers, double *partial, long long partial_stride,
                                   double *perstar, int tiles_per_block, int n_groups, int heavy_parts,
                                   hipStream_t stream)
{
    // + 8: find_bracket's last stage may read up to 6 entries past a column's end (masked out)
    const size_t lds = sizeof(double) * std::max((size_t)WB * NPOPS * mass_cap + 8, heavy_lds_doubles(pk, NPOPS));
    auto kern = k_star_like<NFP, NPOPS, WB>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int n_wsets = (n_walkers + WB - 1) / WB;
    const int hot = 8 * ((n_groups + 7) / 8) * n_wsets;          // padded so every XCD sees whole walker sets
    const int heavy = (n_walkers * heavy_parts + 7) / 8 * 8;     // heavy-star workgroups lead the grid
    hipLaunchKernelGGL(kern, dim3(heavy + hot), dim3(256), lds, stream, pk, st, hdr, iso_data,
                       iso_stride, mass_cap, d_params, n_walkers, partial, partial_stride, n_groups, perstar,
                       tiles_per_block, heavy, heavy_parts);
    return hipGetLastError();
}

#define B9_SWITCH_NFP(CALL2, CALL1)                 \
    switch (pk.nfp) {                               \
    case 4:  if (n_pops == 2) { return CALL2(4); } else { return CALL1(4); }   \
    case 8:  if (n_pops == 2) { return CALL2(8); } else { return CALL1(8); }   \
    case 16: if (n_pops == 2) { return CALL2(16); } else { return CALL1(16); } \
    default: return hipErrorInvalidValue;           \
    }

hipError_t b9k_star_like(const DevPack &pk, const DevStars &st, const IsoHdr *hdr,
                         const double *iso_data, long long iso_stride, int mass_cap,
                         const double *d_params, int n_walkers, int n_pops, int wb,
                         double *partial, long long partial_stride, double *perstar, int tiles_per_block,
                         int n_groups, int heavy_parts, hipStream_t stream)
{
#define SL_ARGS pk, st, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, partial, partial_stride, perstar, tiles_per_block, n_groups, heavy_parts, stream
#define SL2(NFP) (wb >= 2 ? launch_star_like<NFP, 2, 2>(SL_ARGS) : launch_star_like<NFP, 2, 1>(SL_ARGS))
#define SL1(NFP) (wb >= 2 ? launch_star_like<NFP, 1, 2>(SL_ARGS) : launch_star_like<NFP, 1, 1>(SL_ARGS))
    B9_SWITCH_NFP(SL2, SL1)
#undef SL1
#undef SL2
#undef SL_ARGS
}

hipError_t b9k_finalize(const IsoHdr *hdr, const double *partial, int n_partial, long long partial_stride,
                        int n_pops, const double *d_params, const DevPriors &pr, int n_walkers, double *d_logpost,
                        double *perstar, int n_stars, const McmcDev &mc, hipStream_t stream)
{
    hipLaunchKernelGGL(k_finalize, dim3(n_walkers), dim3(256), 0, stream, hdr, partial, n_partial, partial_stride,
                       n_pops, d_params, pr, d_logpost, perstar, n_stars, mc);
    return hipGetLastError();
}

template <int NFP, int NPOPS, bool SAMPLE>
static hipError_t launch_star_marg_t(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                                     long long iso_stride, int mass_cap, const double *d_params, int n_walkers,
                                     double *vals, double *perstar, int K, int Q, const B9MargSample *smp, hipStream_t stream)
{
    size_t lds = sizeof(double) * (size_t)NPOPS * mass_cap * (NFP + 1);
    int chunk_cap = ((mass_cap - 1) * K + 63) / 64 + 1;                     // chunk-bound table, when LDS has room for it
    // three planes + per wave: 2 NFP doubles of the star's offsets/weights and a list of surviving chunks
    const size_t with_table = lds + sizeof(double) * ((size_t)NPOPS * chunk_cap * NFP * 3 + 4 * 2 * NFP) + sizeof(int) * 4 * ((size_t)chunk_cap + 1);
    if (with_table <= 160 * 1024 && !getenv("B9_NO_CHUNK_BOUNDS")) lds = with_table; else chunk_cap = 0;
    auto kern = k_star_marg<NFP, NPOPS, SAMPLE>;
    MargSample ms{};
    if (SAMPLE) { ms.mass = smp->mass; ms.ratio = smp->ratio; ms.member = smp->member; ms.pop = smp->pop; ms.k0 = smp->k0; ms.k1 = smp->k1; ms.row0 = smp->row0; }
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((st.n_pad + 3) / 4, n_walkers), dim3(256), lds, stream, pk, st, hdr, iso_data, iso_stride,
                       mass_cap, d_params, vals, perstar, K, Q, ms, chunk_cap);
    return hipGetLastError();
}

template <int NFP, int NPOPS>
static hipError_t launch_star_marg(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                                   long long iso_stride, int mass_cap, const double *d_params, int n_walkers,
                                   double *vals, double *perstar, int K, int Q, const B9MargSample *smp, hipStream_t stream)
{
    return smp ? launch_star_marg_t<NFP, NPOPS, true>(pk, st, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, vals, perstar, K, Q, smp, stream)
               : launch_star_marg_t<NFP, NPOPS, false>(pk, st, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, vals, perstar, K, Q, smp, stream);
}

hipError_t b9k_star_marg(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                         long long iso_stride, int mass_cap, const double *d_params, int n_walkers, int n_pops,
                         double *vals, double *perstar, int K, int Q, const B9MargSample *smp, hipStream_t stream)
{
#define SM_ARGS pk, st, hdr, iso_data, iso_stride, mass_cap, d_params, n_walkers, vals, perstar, K, Q, smp, stream
#define SM2(NFP) launch_star_marg<NFP, 2>(SM_ARGS)
#define SM1(NFP) launch_star_marg<NFP, 1>(SM_ARGS)
    B9_SWITCH_NFP(SM2, SM1)
#undef SM1
#undef SM2
#undef SM_ARGS
}

template <int NFP, int NPOPS>
static hipError_t launch_mcmc_step(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr,
                                   int tiles_per_block, int n_groups, int heavy_parts, int derive_parts, int derive_order, hipStream_t stream)
{
    // hot role: the mass columns of both candidates (+ 8: find_bracket's masked over-read)
    const size_t stage = B9_LDS_STAGE(NFP) ? (size_t)4 * (2 * NFP + 1) * 64 : 0;      // per-wave observation stage of the hot role
    const size_t lds = sizeof(double) * std::max((size_t)2 * NPOPS * sd.mass_cap + 8 + stage, heavy_lds_doubles(pk, NPOPS));
    auto kern = k_mcmc_step<NFP, NPOPS>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int W = sd.n_walkers;
    const int hot = 8 * ((n_groups + 7) / 8) * W;
    const int derive_first = derive_order >= 0 ? 1 : 0;
    const int n_derive = W * 2 * NPOPS * derive_parts;
    const int front = (W * heavy_parts + (derive_first ? n_derive : W) + 7) / 8 * 8;
    const int back = (!derive_first && sd.derive_next) ? n_derive : 0;
    hipLaunchKernelGGL(kern, dim3(front + hot + back), dim3(256), lds, stream, pk, st, sd, pr, tiles_per_block, n_groups,
                       front, hot, heavy_parts, derive_parts, derive_first);
    return hipGetLastError();
}

hipError_t b9k_mcmc_step(const DevPack &pk, const DevStars &st, const StepDev &sd, const DevPriors &pr,
                         int tiles_per_block, int n_groups, int heavy_parts, int derive_parts, int derive_order, hipStream_t stream)
{
    const int n_pops = sd.n_pops;
#define MS_ARGS pk, st, sd, pr, tiles_per_block, n_groups, heavy_parts, derive_parts, derive_order, stream
#define MS2(NFP) launch_mcmc_step<NFP, 2>(MS_ARGS)
#define MS1(NFP) launch_mcmc_step<NFP, 1>(MS_ARGS)
    B9_SWITCH_NFP(MS2, MS1)
#undef MS1
#undef MS2
#undef MS_ARGS
}

hipError_t b9k_mcmc_finish(const DevPack &pk, const StepDev &sd, const DevPriors &pr, hipStream_t stream)
{
    hipLaunchKernelGGL(k_mcmc_finish, dim3(sd.n_walkers), dim3(256), 0, stream, pk, sd, pr);
    return hipGetLastError();
}

// An empty kernel: bracketing it with HIP events measures what an event bracket adds to a kernel's
// own duration (dispatch boundary + event processing); b9_calibrate_timing subtracts nothing by
// itself, it only reports the figure.
__global__ void k_noop(int *p) { if (p && threadIdx.x == 1024) *p = 0; }
// keeps the queue busy for ~`ticks` s_memrealtime ticks (100 MHz) so that work enqueued behind it
// executes back to back (bounded spin; one wave)
__global__ void k_spin(unsigned long long ticks, int *p)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int guard = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && guard < (1 << 24)) ++guard;
    if (p && guard < 0) *p = guard;
}
hipError_t b9k_spin(double microseconds, hipStream_t stream)
{
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, stream, (unsigned long long)(microseconds * 100.0), (int *)nullptr);
    return hipGetLastError();
}
hipError_t b9k_noop(hipStream_t stream)
{
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, stream, (int *)nullptr);
    return hipGetLastError();
}
