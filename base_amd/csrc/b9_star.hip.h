// b9_star.hip.h -- general per-star evolution: MS/RGB lookup, IFMR, WD cooling + atmosphere, system chi^2 (rows a4-a7, a9).
// Part of the single translation unit b9_kernels.hip (included there, in this order); gfx950 only.
#pragma once

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

// num / den for the heavy-star role's interpolation weights: a v_rcp_f64 seed, two Newton steps and a residual
// correction -- within 1 ulp of the IEEE quotient (weights then differ from the oracle's by <= 1e-16 relative, seven
// orders inside the stated tolerance) in a third of the dependent instructions of the exact division sequence.  The
// role is one long dependent chain: every instruction on it is latency.
__device__ __forceinline__ double fdiv(double num, double den)
{
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    const double q = num * r;
    return fma(fma(-den, q, num), r, q);
}

// Bracket of mass m in a mass column (LDS-resident in the hot roles; any pointer works): the largest i in [0, n-2] with mass[i] <= m
// (what the oracle's binary search returns -- the bracket is unique for a sorted column, so any
// correct search yields the same i and hence bit-identical weights).  8-ary: every step issues 7
// independent ds_reads and narrows the range eightfold, so a 400-point column takes 3 dependent
// LDS round trips instead of the 9 of a binary search (measured: the binary search was 19 % of the
// kernel's VALU instructions but 3.3 of its 20.5 us).
// The index alone, for an ascending (DESC = false: largest i <= n-2 with ax[i] <= x) or descending (DESC = true:
// largest i <= n-2 with ax[i] >= x) axis; equal to the oracle's binary searches, which return the same unique index.
// Probes of a bracket search.  The N probes of a round are independent loads whose answers are needed together; left
// to itself the compiler (a) turns `j < len && p[j] <= x` into seven branches with a load and a wait each, and (b)
// under register pressure schedules unconditional probes one at a time through one register pair -- seven dependent
// LDS round trips where one was meant (measured: a WD star's chain of ~10 searches took 7 us of its 9.7).  So: the
// loads are unconditional (clamped index, never past the axis), written into an array before any is looked at, and
// fenced so the scheduler keeps them together (B9_PROBE_FENCE).
#define B9_PROBE_FENCE() __builtin_amdgcn_sched_barrier(0)
template <bool DESC>
__device__ __forceinline__ int count7(const double (&v)[7], double x)
{
    int c = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) c += (DESC ? v[j] >= x : v[j] <= x) ? 1 : 0;
    return c;
}
// values at p[step], p[2 step], ... p[7 step]
__device__ __forceinline__ void load7(const double *p, int step, double (&v)[7])
{
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = p[(j + 1) * step];
}
// values at p[1] ... p[7], those at or past `len` replaced by p[0]'s (never reads past the axis)
__device__ __forceinline__ void load7_tail(const double *p, int len, double (&v)[7])
{
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = p[(j + 1) < len ? (j + 1) : 0];
}
template <bool DESC>
__device__ __forceinline__ int count7_tail(const double (&v)[7], int len, double x)
{
    int c = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) c += (int)((j + 1) < len) & (int)(DESC ? v[j] >= x : v[j] <= x);
    return c;
}

template <bool DESC>
__device__ __forceinline__ int bracket8(const double *ax, int n, double x)
{
    int lo = 0, len = n - 1;                 // the answer lies in [lo, lo + len)
    if (n < 2) return 0;
    double v[7];
    while (len >= 8) {                       // 7 probes at lo + j*step, all inside the range (7*step < len)
        const int step = len >> 3;
        B9_PROBE_FENCE();
        load7(ax + lo, step, v);
        B9_PROBE_FENCE();
        const int c = count7<DESC>(v, x);
        lo += c * step;
        len = (c == 7) ? len - 7 * step : step;
    }
    B9_PROBE_FENCE();                        // fewer than 8 candidates left: probe them all at once
    load7_tail(ax + lo, len, v);
    B9_PROBE_FENCE();
    return lo + count7_tail<DESC>(v, len, x);
}

__device__ __forceinline__ void find_bracket(const double *mass, int n, double m, int &lo_out, double &t_out)
{
    int lo = 0, len = n - 1;                 // the answer lies in [lo, lo + len)
    double v[7];
    while (len >= 8) {                       // 7 probes at lo + j*step, all inside the range (7*step < len)
        const int step = len >> 3;
        B9_PROBE_FENCE();
        load7(mass + lo, step, v);
        B9_PROBE_FENCE();
        const int c = count7<false>(v, m);
        lo += c * step;
        len = (c == 7) ? len - 7 * step : step;
    }
    {                                        // fewer than 8 candidates left: probe them all at once
        B9_PROBE_FENCE();
        load7(mass + lo, 1, v);              // reads stay inside the column: lo + 7 <= n + 6 < capacity
        B9_PROBE_FENCE();
        lo += count7_tail<false>(v, len, m);
    }
    const double a = mass[lo], d = mass[lo + 1] - a;
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
    lo_out = lo;
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
    const double *wc_mass, *wc_carb, *at_log_teff, *at_logg;
    const double *wc_log_age_lds; // LDS copy of the cooling tracks' concatenated age axes, or null (then pk.wc_log_age, in L2, is searched)
    const double *wc_track;       // per track: (points | first point << 32) packed in the bits of a double (DevPack::wc_track, or its LDS copy)
};

__device__ inline double prec_log_age_corner(const DevPack &pk, const double *tips, const double *log_age, double m)
{
    const int na = pk.n_age;
    const double tip0 = tips[0];
    if (m > tip0) return log_age[0] - 2.7 * log10(m / tip0);
    if (m <= tips[na - 1]) return log_age[na - 1];
    const int lo = bracket8<true>(tips, na, m);              // (here tips[0] >= m > tips[na-1])
    const double a = tips[lo], b = tips[lo + 1];
    const double t = (b != a) ? (m - a) / (b - a) : 0.0;
    return lerp(log_age[lo], log_age[lo + 1], t);
}

// Several 8-ary bracket searches in lock step (one per cooling track): the probes of all N axes are issued together in
// every round, so N searches cost the dependent round trips of the longest one.  Same indices as N calls of
// bracket8<false>.
template <int N>
__device__ __forceinline__ void bracket8_lockstep(const double *const (&ax)[N], const int (&n)[N], double x, int (&lo)[N])
{
    int len[N];
    bool any = false;
#pragma unroll
    for (int k = 0; k < N; ++k) { lo[k] = 0; len[k] = n[k] - 1; any = any || len[k] >= 8; }
    double v[N][7];
    while (any) {
        any = false;
        int step[N];
        B9_PROBE_FENCE();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            step[k] = len[k] >= 8 ? len[k] >> 3 : 0;             // an axis that is done probes its own lo (ignored)
            load7(ax[k] + lo[k], step[k], v[k]);
        }
        B9_PROBE_FENCE();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const bool on = len[k] >= 8;
            const int c = count7<false>(v[k], x);
            lo[k] += on ? c * step[k] : 0;
            len[k] = on ? ((c == 7) ? len[k] - 7 * step[k] : step[k]) : len[k];
            any = any || len[k] >= 8;
        }
    }
    B9_PROBE_FENCE();
#pragma unroll
    for (int k = 0; k < N; ++k) load7_tail(ax[k] + lo[k], len[k], v[k]);
    B9_PROBE_FENCE();
#pragma unroll
    for (int k = 0; k < N; ++k) lo[k] += count7_tail<false>(v[k], len[k], x);
}

// WD cooling model: (log Teff, log radius) of a WD of mass wd_mass at log cooling age log_cool.
// Every (carbonicity, mass) node is a track with ITS OWN age axis: the age is bracketed (clamped; extrapolation
// allowed) in each of the 2 (4 with a carbonicity axis) neighbouring tracks' axes, the two quantities are interpolated
// along each track, then across mass, then across carbonicity.  ax.wc_* may point into LDS.
template <int NT>
__device__ __forceinline__ void wd_cooling_tracks(const DevPack &pk, const WdAxes &ax, int ic, int im, double tm, double tc,
                                                  double log_cool, const double *age_base, double &log_teff, double &log_rad)
{
    const double *axes[NT];
    int n[NT], off[NT], ia[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {                               // k = dc * 2 + dm
        const int t = (ic + (k >> 1)) * pk.n_wc_mass + im + (k & 1);
        const unsigned long long w = __double_as_longlong(ax.wc_track[t]);
        n[k] = (int)(w & 0xFFFFFFFFull); off[k] = (int)(w >> 32);
        axes[k] = age_base + off[k];
    }
    HS2(13);
    double vt[NT], vr[NT];
    if (pk.wc_uniform) {          // a rectangular table: the tracks share one age axis -- one search, one weight for all of them
        const int i0 = bracket8<false>(age_base, n[0], log_cool);
        const double a0 = age_base[i0], a1 = age_base[i0 + 1];
        const double ta = fdiv(log_cool - a0, a1 - a0);
        HS2(14);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const size_t b = (size_t)off[k] + i0;
            vt[k] = lerp(pk.wc_log_teff[b], pk.wc_log_teff[b + 1], ta);
            vr[k] = lerp(pk.wc_log_radius[b], pk.wc_log_radius[b + 1], ta);
        }
    } else {
        if constexpr (NT == 4) {
            // one track after the other: four columns' probes at once are 56 registers, the peak of the whole role (it
            // spilled there); a pack with a carbonicity axis pays the extra LDS search rounds on its WD stars
#pragma unroll 1
            for (int k = 0; k < NT; ++k) {
                const double *a = (k & 2) ? ((k & 1) ? axes[NT - 1] : axes[NT - 2]) : ((k & 1) ? axes[1] : axes[0]);
                const int nk = (k & 2) ? ((k & 1) ? n[NT - 1] : n[NT - 2]) : ((k & 1) ? n[1] : n[0]);
                const int r = bracket8<false>(a, nk, log_cool);
                if (k == 0) ia[0] = r; else if (k == 1) ia[1] = r; else if (k == 2) ia[NT - 2] = r; else ia[NT - 1] = r;
            }
        } else {
            bracket8_lockstep<NT>(axes, n, log_cool, ia);
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const double a0 = axes[k][ia[k]], a1 = axes[k][ia[k] + 1];
            const double ta = fdiv(log_cool - a0, a1 - a0);
            const size_t b = (size_t)off[k] + ia[k];
            vt[k] = lerp(pk.wc_log_teff[b], pk.wc_log_teff[b + 1], ta);
            vr[k] = lerp(pk.wc_log_radius[b], pk.wc_log_radius[b + 1], ta);
        }
    }
    log_teff = lerp(vt[0], vt[1], tm);
    log_rad = lerp(vr[0], vr[1], tm);
    if (NT == 4) {
        log_teff = lerp(log_teff, lerp(vt[NT - 2], vt[NT - 1], tm), tc);
        log_rad = lerp(log_rad, lerp(vr[NT - 2], vr[NT - 1], tm), tc);
    }
}

__device__ __forceinline__ void wd_cooling(const DevPack &pk, const WdAxes &ax, const double *__restrict__ par, double wd_mass,
                                           double log_cool, double &log_teff, double &log_rad)
{
    const int im = bracket8<false>(ax.wc_mass, pk.n_wc_mass, wd_mass);
    const double tm = fdiv(wd_mass - ax.wc_mass[im], ax.wc_mass[im + 1] - ax.wc_mass[im]);
    if (pk.n_wc_carb > 1) {
        const int ic = bracket8<false>(ax.wc_carb, pk.n_wc_carb, par[B9_P_CARBONICITY]);
        const double tc = fdiv(par[B9_P_CARBONICITY] - ax.wc_carb[ic], ax.wc_carb[ic + 1] - ax.wc_carb[ic]);
        // (two calls, not one with a selected pointer: each keeps its address space -- ds_read for the LDS copy)
        // (a rectangular table: the LDS copy holds the one shared axis; the global one starts at point wc_off0)
        const double *g = pk.wc_log_age + (pk.wc_uniform ? pk.wc_off0 : 0);
        if (ax.wc_log_age_lds) wd_cooling_tracks<4>(pk, ax, ic, im, tm, tc, log_cool, ax.wc_log_age_lds, log_teff, log_rad);
        else wd_cooling_tracks<4>(pk, ax, ic, im, tm, tc, log_cool, g, log_teff, log_rad);
    } else {
        const double *g = pk.wc_log_age + (pk.wc_uniform ? pk.wc_off0 : 0);
        if (ax.wc_log_age_lds) wd_cooling_tracks<2>(pk, ax, 0, im, tm, 0.0, log_cool, ax.wc_log_age_lds, log_teff, log_rad);
        else wd_cooling_tracks<2>(pk, ax, 0, im, tm, 0.0, log_cool, g, log_teff, log_rad);
    }
}

// Precursor log-age of a WD progenitor of mass m: prec_log_age_corner for the NC (2, or 4 with a helium axis) corner
// columns of AGB-tip masses IN LOCK STEP -- the probes of all columns are issued together in every round of the
// descending 8-ary search, so NC inversions cost the dependent round trips of one -- then interpolated in Y and FeH.
// Same operations per corner as prec_log_age_corner (same bits).
template <int NC>
__device__ __forceinline__ void prec_corners(const DevPack &pk, const double *const (&tips)[4], const double *log_age, double m, double (&out)[4])
{
    const int na = pk.n_age;
    int lo[NC], len[NC];
    bool heavy[NC], light[NC], any = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        heavy[c] = m > tips[c][0];
        light[c] = m <= tips[c][na - 1];
        lo[c] = 0; len[c] = (heavy[c] || light[c]) ? 0 : na - 1;
        any = any || len[c] >= 8;
    }
    double v[NC][7];
    while (any) {
        any = false;
        int step[NC];
        B9_PROBE_FENCE();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            step[c] = len[c] >= 8 ? len[c] >> 3 : 0;
            load7(tips[c] + lo[c], step[c], v[c]);
        }
        B9_PROBE_FENCE();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool on = len[c] >= 8;
            const int k = count7<true>(v[c], m);
            lo[c] += on ? k * step[c] : 0;
            len[c] = on ? ((k == 7) ? len[c] - 7 * step[c] : step[c]) : len[c];
            any = any || len[c] >= 8;
        }
    }
    B9_PROBE_FENCE();
#pragma unroll
    for (int c = 0; c < NC; ++c) load7_tail(tips[c] + lo[c], len[c], v[c]);
    B9_PROBE_FENCE();
#pragma unroll
    for (int c = 0; c < NC; ++c) lo[c] += count7_tail<true>(v[c], len[c], m);
    double ta[NC], tb[NC], la[NC], lb[NC];
    B9_PROBE_FENCE();
#pragma unroll
    for (int c = 0; c < NC; ++c) { ta[c] = tips[c][lo[c]]; tb[c] = tips[c][lo[c] + 1]; la[c] = log_age[lo[c]]; lb[c] = log_age[lo[c] + 1]; }
    B9_PROBE_FENCE();
    const double la_last = log_age[na - 1];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double t = (tb[c] != ta[c]) ? fdiv(m - ta[c], tb[c] - ta[c]) : 0.0;
        const double val = lerp(la[c], lb[c], t);
        out[c] = light[c] ? la_last : val;
    }
    // heavier than a column's youngest tip: extrapolated (rare; the logarithm is only paid when some lane needs it)
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (heavy[c]) out[c] = log_age[0] - 2.7 * log10(m / tips[c][0]);
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
    // (fully unrolled: a run-time index into ax.tips[] would put the pointer array in scratch memory)
    const double v00 = prec_log_age_corner(pk, ax.tips[0], ax.log_age, m), v10 = prec_log_age_corner(pk, ax.tips[2], ax.log_age, m);
    double vf0 = v00, vf1 = v10;
    if (ny == 2) {
        vf0 = lerp(v00, prec_log_age_corner(pk, ax.tips[1], ax.log_age, m), iso.t_y);
        vf1 = lerp(v10, prec_log_age_corner(pk, ax.tips[3], ax.log_age, m), iso.t_y);
    }
    const double prec = lerp(vf0, vf1, iso.t_feh);
    const double log_age = par[B9_P_LOGAGE];
    if (prec >= log_age) { fill<NFP>(out, -4.0); return; }
    const double wd_mass = ifmr(pk.ifmr_id, par, m);
    const double log_cool = log10(exp10(log_age) - exp10(prec));

    double tr[2];
    wd_cooling(pk, ax, par, wd_mass, log_cool, tr[0], tr[1]);
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
        const double d = pred - st.obs[B9_SIDX(NFP, f, i)];
        chi2 = fma(st.w[B9_SIDX(NFP, f, i)] * d, d, chi2);
    }
    // a non-finite predicted magnitude (NaN or inf, also under a zero weight: 0 * inf = NaN)
    // leaves chi2 non-finite: the star is impossible under this isochrone
    return isfinite(chi2) ? chi2 : __builtin_inf();
}

// ------------------------------------------------------------------------------------------
// Lean form of the general per-star evolution, used by the heavy-star role (stars above a walker's AGB tip).
// A component (primary or secondary) is first reduced to a DESCRIPTOR of what its magnitudes are interpolated
// from -- nothing, one pair of isochrone rows, four atmosphere rows, or a constant -- and the filters are then
// walked once, each magnitude formed where it is used.  Same operations in the same order as star_mags /
// chi2_system above (identical bits), but no per-filter arrays: the role needs a third of the registers, so it
// neither spills nor sets the register budget of the kernel it shares with the hot role.
// ------------------------------------------------------------------------------------------
#define B9_HEAVY_UNROLL 4          // filters per batch of table-row loads (a batch = one memory round trip; 8 costs 15 more VGPRs)
struct Comp {
    int kind;               // 0: no flux (99.999)   1: two rows, lerp t   2: four rows, lerp t then tg   3: constant -4
    const double *r0, *r1;  // kind 1: r0 = lower isochrone row (upper = r0 + NFP); kind 2: r0 / r1 = the two log g rows
    double t, tg;
};

template <int NFP>
__device__ __forceinline__ Comp comp_desc(const DevPack &pk, const WdAxes &ax, const double *is_mass, const double *is_mags,
                                          int is_n, double is_tip, double t_feh, double t_y,
                                          const double *__restrict__ par, double m, int wd_type)
{
    Comp c; c.kind = 0; c.r0 = is_mags; c.r1 = is_mags; c.t = 0.0; c.tg = 0.0;
    if (!(m > 0.0)) return c;
    if (m <= is_tip) {                                           // MS / RGB (msrgb_mags)
        if (m < is_mass[0]) return c;
        int lo; double t;
        HS2(11);
        find_bracket(is_mass, is_n, m, lo, t);
        HS2(12);
        c.kind = 1; c.r0 = is_mags + (size_t)lo * NFP; c.t = t;
        return c;
    }
    if (!(m <= pk.m_wd_up)) return c;                            // NS / BH
    if (pk.n_wc_mass < 2 || pk.n_at_teff < 2) return c;          // no WD models loaded
    // WD (wd_mags): precursor age -> cooling age -> (Teff, radius) -> atmosphere rows
    HS2(3);
    double pc[4];
    double vf0, vf1;
    if (pk.n_y > 1) {
        prec_corners<4>(pk, ax.tips, ax.log_age, m, pc);
        vf0 = lerp(pc[0], pc[1], t_y); vf1 = lerp(pc[2], pc[3], t_y);
    } else {
        const double *const two[4] = {ax.tips[0], ax.tips[2], ax.tips[0], ax.tips[2]};
        prec_corners<2>(pk, two, ax.log_age, m, pc);
        vf0 = pc[0]; vf1 = pc[1];
    }
    HS2(2); B9_MARK("hv-prec-done");
    const double prec = lerp(vf0, vf1, t_feh);
    const double log_age = par[B9_P_LOGAGE];
    if (prec >= log_age) { c.kind = 3; return c; }
    const double wd_mass = ifmr(pk.ifmr_id, par, m);
    const double log_cool = log10(exp10(log_age) - exp10(prec));
    HS2(4); B9_MARK("hv-explog-done");
    double tr[2];
    wd_cooling(pk, ax, par, wd_mass, log_cool, tr[0], tr[1]);
    HS2(6); B9_MARK("hv-cooling-done");
    const double log_teff = tr[0];
    const double logg = LOG_G_PLUS_LOG_MSUN + log10(wd_mass) - 2.0 * tr[1];
    const int ty = (wd_type > 0 && pk.n_at_type > 1) ? 1 : 0;
    const int it = bracket8<false>(ax.at_log_teff, pk.n_at_teff, log_teff);
    c.t = fdiv(log_teff - ax.at_log_teff[it], ax.at_log_teff[it + 1] - ax.at_log_teff[it]);
    const int ig = bracket8<false>(ax.at_logg, pk.n_at_logg, logg);
    c.tg = fdiv(logg - ax.at_logg[ig], ax.at_logg[ig + 1] - ax.at_logg[ig]);
    c.r0 = pk.at_mags + (((size_t)ty * pk.n_at_logg + ig) * pk.n_at_teff + it) * NFP;
    c.r1 = c.r0 + (size_t)pk.n_at_teff * NFP;
    c.kind = 2;
    HS2(7); B9_MARK("hv-desc-done");
    return c;
}

// Magnitude of a component in filter f.  Branch-free on purpose: the four table words are requested whatever the
// kind (kinds 0, 1 and 3 point r0 / r1 at valid rows and ignore what comes back), so the loads of a whole batch of
// filters -- both components, every population -- are in flight together; under per-kind branches every filter
// paid its own memory round trip.  The selected value is formed by exactly star_mags' operations.
template <int NFP>
__device__ __forceinline__ double comp_mag(const Comp &c, int f)
{
    const double a = c.r0[f], b = c.r0[NFP + f], e = c.r1[f], g = c.r1[NFP + f];
    const double v0 = lerp(a, b, c.t), v1 = lerp(e, g, c.t);
    const double two = lerp(v0, v1, c.tg);
    return c.kind == 1 ? v0 : (c.kind == 2 ? two : (c.kind == 3 ? -4.0 : B9_MAG_NOFLUX));
}

// What one lane of the heavy-star role works on: ONE component of a system in ONE population under ONE candidate
// parameter row -- the caller has selected these for the lane (field by field: indexing local arrays of views with a
// per-lane index would put them in scratch memory).
template <int NFP>
struct LaneView {
    const double *is_mass, *is_mags;  // the (candidate, population)'s derived isochrone
    int is_n;
    double is_tip, t_feh, t_y;
    const double *par;                // the candidate's parameter row
    WdAxes ax;                        // WD axes (LDS) with the (candidate, population)'s AGB-tip columns
};

// One star of ONE population through the descriptors, spread over a PAIR of neighbouring lanes: lane `comp` of the pair
// evaluates component `comp`.  The heavy role is a latency chain (a WD descriptor is ~30 dependent LDS search steps, three
// library transcendentals and two memory round trips); laid end to end in one lane a binary cost two of them, side by side
// one.  The pair swaps component descriptors by wave shuffles and then splits the FILTERS (each lane forms both
// components' magnitudes in half of them).  Every magnitude is formed by star_value's operations; the chi^2 is the sum of two
// half-sums instead of one chain over the filters (a 1e-16 relative difference).  Both lanes of a pair must call (full
// EXEC); the result -- the star's log-likelihood in this population, c0 included -- is valid in both.  Two populations are
// two WAVES (heavy_stars), not more lanes: the lane view stays wave-uniform.
// What the HEAD of a heavy star's chain needs (heavy-order arrays, DevStars::hv_*): nothing here depends on the
// candidate, so the caller requests the first chunk's at the role's entry.  Everything else of the star -- the
// observations and weights, its two constants -- is used at the chain's END and requested there, together with the
// table rows (one round trip for all of it): the role is short of registers, not of loads (carrying the 20 values
// through the chain cost spills and 0.5-0.8 us per step).
struct HeavyStar {
    double m1, q;
    int flags;
};

__device__ __forceinline__ HeavyStar load_heavy_star(const DevStars &st, int j /* index in the descending-mass list */)
{
    HeavyStar h;
    h.m1 = st.heavy_mass[j]; h.q = st.hv_q[j]; h.flags = st.hv_flags[j];
    return h;
}

template <int NFP>
__device__ __forceinline__ double star_ll_lanes(const DevPack &pk, const LaneView<NFP> &lv, const DevStars &st,
                                                int j /* index in the descending-mass list */, const HeavyStar &hs, int comp)
{
    constexpr int HF = NFP / 2;                                       // filters per lane of the pair
    const double m1 = hs.m1, q = hs.q;
    const int wd_type = hs.flags & 1;
    const bool binary = q > 0.0;
    HS2(1);
    // 1. each lane of the pair reduces ITS component to a descriptor (the long chain; the two run side by side)
    Comp d; d.kind = 0; d.r0 = lv.is_mags; d.r1 = lv.is_mags; d.t = 0.0; d.tg = 0.0;
    if (comp == 0 || binary)
        d = comp_desc<NFP>(pk, lv.ax, lv.is_mass, lv.is_mags, lv.is_n, lv.is_tip, lv.t_feh, lv.t_y, lv.par, comp ? q * m1 : m1, wd_type);
    // 2. the lanes swap descriptors: both now hold the primary's (c1) and the secondary's (c2) ...
    Comp o;
    o.kind = __shfl_xor(d.kind, 1, 64);
    o.r0 = reinterpret_cast<const double *>(__shfl_xor((unsigned long long)reinterpret_cast<size_t>(d.r0), 1, 64));
    o.r1 = reinterpret_cast<const double *>(__shfl_xor((unsigned long long)reinterpret_cast<size_t>(d.r1), 1, 64));
    o.t = __shfl_xor(d.t, 1, 64); o.tg = __shfl_xor(d.tg, 1, 64);
    Comp c1, c2;
    c1.kind = comp ? o.kind : d.kind; c1.r0 = comp ? o.r0 : d.r0; c1.r1 = comp ? o.r1 : d.r1; c1.t = comp ? o.t : d.t; c1.tg = comp ? o.tg : d.tg;
    c2.kind = comp ? d.kind : o.kind; c2.r0 = comp ? d.r0 : o.r0; c2.r1 = comp ? d.r1 : o.r1; c2.t = comp ? d.t : o.t; c2.tg = comp ? d.tg : o.tg;
    // 3. ... and each takes HALF of the filters (the secondary's lane is not idle through the flux combines; a single
    //    star's second lane, idle until now, takes half of the primary's filters)
    const double c0 = st.hv_c0[j];
    const double mod = lv.par[B9_P_MOD], av = lv.par[B9_P_ABS];      // (LDS; read here, not carried through the chain)
    double obs[HF], wgt[HF];
#pragma unroll
    for (int k = 0; k < HF; ++k) {
        const int f = comp * HF + k;
        obs[k] = st.hv_obs[(size_t)f * st.hv_pad + j]; wgt[k] = st.hv_w[(size_t)f * st.hv_pad + j];
    }
    double chi2 = 0.0;
#pragma unroll
    for (int k = 0; k < HF; ++k) {
        const int f = comp * HF + k;
        double p = comp_mag<NFP>(c1, f);
        const double p2 = comp_mag<NFP>(c2, f);
        if (binary) p -= (2.5 / LN10) * log1pexp((-0.4 * LN10) * (p2 - p));
        const double dd = (p + (mod + pk.abs_m1[f] * av)) - obs[k];
        chi2 = fma(wgt[k] * dd, dd, chi2);
    }
    HS2(8);
    chi2 += __shfl_xor(chi2, 1, 64);                                  // (the sum of the two halves: the same bits in both lanes)
    return c0 - 0.5 * (isfinite(chi2) ? chi2 : __builtin_inf());
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

