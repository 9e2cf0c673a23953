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

