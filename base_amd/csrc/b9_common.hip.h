// b9_common.hip.h -- constants, interpolation / transcendental helpers, Philox, cluster prior,
// fixed-order finish of a log-posterior and the Metropolis accept (shared by every kernel).
// Part of the single translation unit b9_kernels.hip (included there, in this order); gfx950 only.
#pragma once

#define LOG_G_PLUS_LOG_MSUN 26.12302173752
#define MF_MU (-1.02)
#define MF_SIGMA 0.67729
#define LN10 2.302585092994045684
#define NEG_INF (-__builtin_inf())

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

// Cross-lane moves of a double without the LDS crossbar (ds_bpermute, what __shfl_* compile to: two per double and step):
// DPP row shifts inside a row of 16 lanes, gfx950's v_permlane16_swap / v_permlane32_swap between rows.  lane_down<O>(v) in
// lane l = v of lane l + O wherever a reduction towards lane 0 reads it (l + O inside the row for O < 16, l in an even
// row for O = 16, l < 32 for O = 32); other lanes hold something unspecified.  Full EXEC.
template <int O>
__device__ __forceinline__ double lane_down(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (O == 32) {
        lo = (int)__builtin_amdgcn_permlane32_swap((unsigned)lo, (unsigned)lo, false, false)[1];
        hi = (int)__builtin_amdgcn_permlane32_swap((unsigned)hi, (unsigned)hi, false, false)[1];
    } else if (O == 16) {
        lo = (int)__builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)lo, false, false)[1];
        hi = (int)__builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)hi, false, false)[1];
    } else {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 | O, 0xf, 0xf, false);      // row_shl:O
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 | O, 0xf, 0xf, false);
    }
    return __hiloint2double(hi, lo);
}

template <int O>
__device__ __forceinline__ int lane_down(int v)
{
    if (O == 32) return (int)__builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false)[1];
    if (O == 16) return (int)__builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false)[1];
    return __builtin_amdgcn_update_dpp(v, v, 0x100 | O, 0xf, 0xf, false);
}

// sum over the wave in lane 0: the tree v[l] += v[l + 32], += v[l + 16], ... += v[l + 1] (as with __shfl_down: same bits)
__device__ __forceinline__ double wave_sum(double v)
{
    v += lane_down<32>(v); v += lane_down<16>(v);
    v += lane_down<8>(v); v += lane_down<4>(v); v += lane_down<2>(v); v += lane_down<1>(v);
    return v;   // valid in lane 0
}

// lane 0's value in every lane, as a scalar
__device__ __forceinline__ double wave_bcast0(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Seven wave sums at once (the tree walk's seven nodes): the SAME tree as wave_sum for each -- v[l] + v[l + 32], then + 16,
// ... + 1, operands at most commuted -- but with the sums packed side by side: level 32 adds two nodes per instruction (one in
// each half of the wave: v_permlane32_swap hands both halves their partner at once), level 16 four (one per row of 16 lanes,
// v_permlane16_swap), and the levels inside a row run on two registers instead of seven: 42 instructions for 126.
__device__ __forceinline__ void lane_swap32(double &x, double &y)      // x' = [x.lo32, y.lo32], y' = [x.hi32, y.hi32] (halves of the wave)
{
    unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x), yl = (unsigned)__double2loint(y), yh = (unsigned)__double2hiint(y);
    const auto a = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
    x = __hiloint2double((int)b[0], (int)a[0]); y = __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ void lane_swap16(double &x, double &y)      // x' = rows [x0, y0, x2, y2], y' = rows [x1, y1, x3, y3]
{
    unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x), yl = (unsigned)__double2loint(y), yh = (unsigned)__double2hiint(y);
    const auto a = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
    x = __hiloint2double((int)b[0], (int)a[0]); y = __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double lane_read(double v, int lane /* constant */)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_sum7(const double (&a)[7], double (&S)[7])        // S[n]: wave-uniform
{
    double x, y, p01, p23, p45, p6;
    x = a[0]; y = a[1]; lane_swap32(x, y); p01 = x + y;          // lower half: a0[l] + a0[l + 32]; upper: a1[l - 32] + a1[l]
    x = a[2]; y = a[3]; lane_swap32(x, y); p23 = x + y;
    x = a[4]; y = a[5]; lane_swap32(x, y); p45 = x + y;
    x = a[6]; y = a[6]; lane_swap32(x, y); p6 = x + y;
    lane_swap16(p01, p23); double q0 = p01 + p23;                 // rows: node 0, 2, 1, 3  (each: its level-32 sums l + (l + 16))
    lane_swap16(p45, p6);  double q1 = p45 + p6;                  // rows: node 4, 6, 5, -
    q0 += lane_down<8>(q0); q1 += lane_down<8>(q1);
    q0 += lane_down<4>(q0); q1 += lane_down<4>(q1);
    q0 += lane_down<2>(q0); q1 += lane_down<2>(q1);
    q0 += lane_down<1>(q0); q1 += lane_down<1>(q1);
    S[0] = lane_read(q0, 0); S[2] = lane_read(q0, 16); S[1] = lane_read(q0, 32); S[3] = lane_read(q0, 48);
    S[4] = lane_read(q1, 0); S[6] = lane_read(q1, 16); S[5] = lane_read(q1, 32);
}

// maximum over the wave, in every lane (wave-uniform; exact whatever the order)
__device__ __forceinline__ double wave_max_all(double v)
{
    double t;
    t = lane_down<32>(v); v = t > v ? t : v;
    t = lane_down<16>(v); v = t > v ? t : v;
    t = lane_down<8>(v); v = t > v ? t : v;
    t = lane_down<4>(v); v = t > v ? t : v;
    t = lane_down<2>(v); v = t > v ? t : v;
    t = lane_down<1>(v); v = t > v ? t : v;
    return wave_bcast0(v);
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
    return isfinite(lp) ? lp : NEG_INF;      // a NaN parameter is outside the support, as the oracle treats it
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

// A double that is the same in every lane of the wave, said so: it then lives in a scalar register pair.
__device__ __forceinline__ double wave_uniform(double x)
{
    const unsigned long long b = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
