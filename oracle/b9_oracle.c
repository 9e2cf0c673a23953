/*
 * b9_oracle.c -- CPU fp64 restatement of the BASE-9 per-step log-posterior path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the reported CPU baseline.  The product path (libbase9hip.so) never
 * links, loads or calls it.
 *
 * PARITY UNPINNED.  The reference source is not mounted: /root/reference holds only
 * README.md, whose line 4 redirects to BayesianStellarEvolution/base-cpp (not present, no
 * network).  There is no reference file:line to follow and no golden vector, known-answer
 * test or fixture to pin this restatement against.  Every function therefore cites the
 * SURVEY.md section-8a row it restates and, tagged [RECALL], the upstream routine it is
 * believed to correspond to.  DESIGN.md section "Math" is the normative statement; each
 * unverifiable choice is listed there as a named deviation risk.
 *
 * Plain scalar C99, one star at a time, no vectorisation tricks: this is also the "port"
 * CPU baseline that bench.py times on the GPU box's host cores.
 *
 * Floating-point contract shared with the HIP kernels: every linear interpolation is
 * fma(t, b - a, a); compile with -ffp-contract=off so nothing else is fused.  With that,
 * isochrone derivation is bit-exact between this file and the GPU; only exp/log/pow differ.
 */
#include "../include/base9_hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define B9O_LOG_G_PLUS_LOG_MSUN 26.12302173752 /* log10(G * Msun), cgs ([RECALL] constants.hpp) */
#define B9O_MF_MU (-1.02)                       /* Miller-Scalo log-normal IMF ([RECALL])        */
#define B9O_MF_SIGMA 0.67729
#define B9O_LN10 2.302585092994045684

typedef struct b9o_iso {
    int valid;
    int first_eep, n;
    double *mass;     /* [n] */
    double *mags;     /* [n * n_filt] point-major */
    double agb_tip;
    /* grid bracket, reused by the WD precursor-age lookup */
    int i_feh, i_y, i_age;
    double t_feh, t_y, t_age;
} b9o_iso;

static inline double lerp(double a, double b, double t) { return fma(t, b - a, a); }

/* largest i in [0, n-2] with ax[i] <= x (clamped) */
static int bracket(const double *ax, int n, double x)
{
    int lo = 0, hi = n - 1;
    if (n < 2) return 0;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ax[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

static inline int iso_index(const b9_pack *p, int ifeh, int iy, int iage)
{
    return (ifeh * p->n_y + iy) * p->n_age + iage;
}

/*
 * SURVEY 8a row a3 -- [RECALL] MsRgbModel::deriveIsochrone.  Bracket (FeH, Y, logAge) in the
 * grid, intersect the EEP ranges of the (up to) 8 corner isochrones, and interpolate mass and
 * every magnitude EEP-by-EEP: first in age, then in Y, then in FeH.
 */
void b9o_iso_free(b9o_iso *iso)
{
    free(iso->mass); free(iso->mags);
    iso->mass = iso->mags = NULL; iso->n = 0; iso->valid = 0;
}

int b9o_derive_isochrone(const b9_pack *p, double log_age, double feh, double y, b9o_iso *iso)
{
    memset(iso, 0, sizeof *iso);
    const int nf = p->n_filt;
    if (!(log_age >= p->log_age[0] && log_age <= p->log_age[p->n_age - 1])) return 0;
    if (!(feh >= p->feh[0] && feh <= p->feh[p->n_feh - 1])) return 0;
    if (p->n_y > 1 && !(y >= p->y[0] && y <= p->y[p->n_y - 1])) return 0;
    if (p->n_age < 2 || p->n_feh < 2) return 0;

    iso->i_age = bracket(p->log_age, p->n_age, log_age);
    iso->t_age = (log_age - p->log_age[iso->i_age]) / (p->log_age[iso->i_age + 1] - p->log_age[iso->i_age]);
    iso->i_feh = bracket(p->feh, p->n_feh, feh);
    iso->t_feh = (feh - p->feh[iso->i_feh]) / (p->feh[iso->i_feh + 1] - p->feh[iso->i_feh]);
    const int ny = (p->n_y > 1) ? 2 : 1;
    if (p->n_y > 1) {
        iso->i_y = bracket(p->y, p->n_y, y);
        iso->t_y = (y - p->y[iso->i_y]) / (p->y[iso->i_y + 1] - p->y[iso->i_y]);
    } else { iso->i_y = 0; iso->t_y = 0.0; }

    int lo = -2147483647, hi = 2147483647;
    for (int df = 0; df < 2; ++df) for (int dy = 0; dy < ny; ++dy) for (int da = 0; da < 2; ++da) {
        int k = iso_index(p, iso->i_feh + df, iso->i_y + dy, iso->i_age + da);
        int f0 = p->iso_first_eep[k], f1 = f0 + p->iso_n_eep[k];
        if (f0 > lo) lo = f0;
        if (f1 < hi) hi = f1;
    }
    int n = hi - lo;
    if (n < 2) return 0;
    iso->first_eep = lo; iso->n = n;
    iso->mass = (double *)malloc(sizeof(double) * (size_t)n);
    iso->mags = (double *)malloc(sizeof(double) * (size_t)n * (size_t)nf);

    for (int e = 0; e < n; ++e) {
        for (int c = 0; c <= nf; ++c) {         /* c == nf: the mass column */
            double vf[2];
            for (int df = 0; df < 2; ++df) {
                double vy[2];
                for (int dy = 0; dy < ny; ++dy) {
                    double va[2];
                    for (int da = 0; da < 2; ++da) {
                        int k = iso_index(p, iso->i_feh + df, iso->i_y + dy, iso->i_age + da);
                        int64_t pt = p->iso_offset[k] + (lo + e - p->iso_first_eep[k]);
                        va[da] = (c == nf) ? p->mass[pt] : p->mags[pt * nf + c];
                    }
                    vy[dy] = lerp(va[0], va[1], iso->t_age);
                }
                vf[df] = (ny == 2) ? lerp(vy[0], vy[1], iso->t_y) : vy[0];
            }
            double v = lerp(vf[0], vf[1], iso->t_feh);
            if (c == nf) iso->mass[e] = v; else iso->mags[(size_t)e * nf + c] = v;
        }
    }
    iso->agb_tip = iso->mass[n - 1];
    iso->valid = 1;
    return 1;
}

/* SURVEY 8a row a7 -- [RECALL] intlFinalMassReln (ifmr.cpp). */
static double ifmr(const b9_pack *p, const double *par, double m)
{
    switch (p->ifmr_id) {
    case B9_IFMR_WEIDEMANN: {
        static const double mi[7] = {1, 2, 3, 4, 5, 6, 7};
        static const double mf[7] = {0.55, 0.60, 0.68, 0.79, 0.88, 0.95, 1.02};
        int i = bracket(mi, 7, m);
        return lerp(mf[i], mf[i + 1], (m - mi[i]) / (mi[i + 1] - mi[i]));
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

/*
 * SURVEY 8a row a7 -- [RECALL] MsRgbModel::wdPrecLogAge: log10 of the precursor's lifetime.
 * Per corner (FeH, Y): invert the AGB-tip-mass(age) curve at this mass; then interpolate the
 * corner values in Y and FeH with the isochrone's own weights.
 */
static double agb_tip_of(const b9_pack *p, int k)
{
    return p->mass[p->iso_offset[k] + p->iso_n_eep[k] - 1];
}

static double prec_log_age_corner(const b9_pack *p, int ifeh, int iy, double m)
{
    const int na = p->n_age;
    const int k0 = iso_index(p, ifeh, iy, 0);
    double tip0 = agb_tip_of(p, k0);
    if (m > tip0)                      /* heavier than the youngest isochrone's tip: t ~ M^-2.7 */
        return p->log_age[0] - 2.7 * log10(m / tip0);
    if (m <= agb_tip_of(p, k0 + na - 1)) return p->log_age[na - 1];
    /* tips descend with age: largest j with tip[j] >= m */
    int lo = 0, hi = na - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (agb_tip_of(p, k0 + mid) >= m) lo = mid; else hi = mid;
    }
    double a = agb_tip_of(p, k0 + lo), b = agb_tip_of(p, k0 + lo + 1);
    double t = (b != a) ? (m - a) / (b - a) : 0.0;
    return lerp(p->log_age[lo], p->log_age[lo + 1], t);
}

static double wd_prec_log_age(const b9_pack *p, const b9o_iso *iso, double m)
{
    const int ny = (p->n_y > 1) ? 2 : 1;
    double vf[2];
    for (int df = 0; df < 2; ++df) {
        double vy[2];
        for (int dy = 0; dy < ny; ++dy)
            vy[dy] = prec_log_age_corner(p, iso->i_feh + df, iso->i_y + dy, m);
        vf[df] = (ny == 2) ? lerp(vy[0], vy[1], iso->t_y) : vy[0];
    }
    return lerp(vf[0], vf[1], iso->t_feh);
}

/*
 * SURVEY 8a row a7 -- [RECALL] Star::wdEvol: IFMR -> cooling model (Teff, radius) ->
 * atmosphere table (Teff, log g) -> magnitudes.
 */
static void wd_mags(const b9_pack *p, const b9o_iso *iso, const double *par, double m,
                    int wd_type, double *out)
{
    const int nf = p->n_filt;
    if (p->n_wc_mass < 2 || p->n_at_teff < 2) { for (int f = 0; f < nf; ++f) out[f] = B9_MAG_NOFLUX; return; }
    double prec = wd_prec_log_age(p, iso, m);
    double log_age = par[B9_P_LOGAGE];
    if (prec >= log_age) { for (int f = 0; f < nf; ++f) out[f] = -4.0; return; } /* tip of the RGB */
    double wd_mass = ifmr(p, par, m);
    double log_cool = log10(pow(10.0, log_age) - pow(10.0, prec));

    /* cooling model: every (carbonicity, mass) node is a track with ITS OWN cooling-age axis
     * (include/base9_hip.h, b9_pack: wc_n_age / wc_offset).  The age is bracketed (clamped; extrapolation allowed) in
     * each neighbouring track's axis and both quantities are interpolated along the track; then across mass, then
     * across carbonicity. */
    int im = bracket(p->wc_mass, p->n_wc_mass, wd_mass);
    double tm = (wd_mass - p->wc_mass[im]) / (p->wc_mass[im + 1] - p->wc_mass[im]);
    int nc = (p->n_wc_carb > 1) ? 2 : 1, ic = 0; double tc = 0.0;
    if (nc == 2) {
        ic = bracket(p->wc_carb, p->n_wc_carb, par[B9_P_CARBONICITY]);
        tc = (par[B9_P_CARBONICITY] - p->wc_carb[ic]) / (p->wc_carb[ic + 1] - p->wc_carb[ic]);
    }
    double tr[2] = {0, 0};   /* log Teff, log radius */
    for (int q = 0; q < 2; ++q) {
        const double *tab = q ? p->wc_log_radius : p->wc_log_teff;
        double vc[2];
        for (int dc = 0; dc < nc; ++dc) {
            double vm[2];
            for (int dm = 0; dm < 2; ++dm) {
                const int t = (ic + dc) * p->n_wc_mass + (im + dm);
                const double *age = p->wc_log_age + p->wc_offset[t];
                const int ia = bracket(age, p->wc_n_age[t], log_cool);
                const double ta = (log_cool - age[ia]) / (age[ia + 1] - age[ia]);
                const size_t base = (size_t)p->wc_offset[t] + ia;
                vm[dm] = lerp(tab[base], tab[base + 1], ta);
            }
            vc[dc] = lerp(vm[0], vm[1], tm);
        }
        tr[q] = (nc == 2) ? lerp(vc[0], vc[1], tc) : vc[0];
    }
    double log_teff = tr[0];
    double logg = B9O_LOG_G_PLUS_LOG_MSUN + log10(wd_mass) - 2.0 * tr[1];

    int ty = (wd_type > 0 && p->n_at_type > 1) ? 1 : 0;
    int it = bracket(p->at_log_teff, p->n_at_teff, log_teff);
    double tt = (log_teff - p->at_log_teff[it]) / (p->at_log_teff[it + 1] - p->at_log_teff[it]);
    int ig = bracket(p->at_logg, p->n_at_logg, logg);
    double tg = (logg - p->at_logg[ig]) / (p->at_logg[ig + 1] - p->at_logg[ig]);
    for (int f = 0; f < nf; ++f) {
        double vg[2];
        for (int dg = 0; dg < 2; ++dg) {
            size_t base = (((size_t)ty * p->n_at_logg + (ig + dg)) * p->n_at_teff + it) * nf + f;
            vg[dg] = lerp(p->at_mags[base], p->at_mags[base + nf], tt);
        }
        out[f] = lerp(vg[0], vg[1], tg);
    }
}

/* SURVEY 8a row a4 -- [RECALL] Star::msRgbEvol: binary search + linear interpolation in mass. */
static void msrgb_mags(const b9_pack *p, const b9o_iso *iso, double m, double *out)
{
    const int nf = p->n_filt;
    if (m < iso->mass[0]) { for (int f = 0; f < nf; ++f) out[f] = B9_MAG_NOFLUX; return; }
    int i = bracket(iso->mass, iso->n, m);
    double d = iso->mass[i + 1] - iso->mass[i];
    double t = (d > 0.0) ? (m - iso->mass[i]) / d : 0.0;
    for (int f = 0; f < nf; ++f)
        out[f] = lerp(iso->mags[(size_t)i * nf + f], iso->mags[(size_t)(i + 1) * nf + f], t);
}

/* [RECALL] Star::getMags / getStatus: which evolutionary branch a ZAMS mass is on. */
static void star_mags(const b9_pack *p, const b9o_iso *iso, const double *par, double m,
                      int wd_type, double *out)
{
    const int nf = p->n_filt;
    if (!(m > 0.0))            { for (int f = 0; f < nf; ++f) out[f] = B9_MAG_NOFLUX; }   /* DNE  */
    else if (m <= iso->agb_tip) msrgb_mags(p, iso, m, out);                               /* MSRG */
    else if (m <= p->m_wd_up)   wd_mags(p, iso, par, m, wd_type, out);                    /* WD   */
    else                       { for (int f = 0; f < nf; ++f) out[f] = B9_MAG_NOFLUX; }   /* NSBH */
}

/*
 * SURVEY 8a row a5 -- [RECALL] StellarSystem::deriveCombinedMags: flux-add primary and
 * secondary, then apply distance modulus and per-filter absorption.  The modulus is (m-M)_V
 * and already contains A_V, hence (coeff_f - 1) * A_V.
 */
static void combined_mags(const b9_pack *p, const b9o_iso *iso, const double *par,
                          double m1, double q, int wd_type, double *out)
{
    const int nf = p->n_filt;
    double m2[64];
    star_mags(p, iso, par, m1, wd_type, out);
    if (q > 0.0) {
        star_mags(p, iso, par, q * m1, wd_type, m2);
        for (int f = 0; f < nf; ++f) {
            double flux = pow(10.0, -0.4 * out[f]) + pow(10.0, -0.4 * m2[f]);
            out[f] = -2.5 * log10(flux);
        }
    }
    for (int f = 0; f < nf; ++f)
        out[f] += par[B9_P_MOD] + (p->abs_coeff[f] - 1.0) * par[B9_P_ABS];
}

static double Phi(double x) { return 0.5 * erfc(-x * M_SQRT1_2); }

/* [RECALL] Cluster::setM_wd_up: normalisation of the IMF over [0.1 Msun, M_wd_up].  The IMF is
 * log-normal in log10(m); per unit mass the density is  c exp(-z^2/2) / (m ln 10)  with
 * z = (log10 m - mu)/sigma, and c makes it integrate to one on the support.  (The closed form
 * recalled from upstream did not integrate to one; DESIGN.md lists this as a deviation risk.) */
double b9o_log_mass_norm(double m_wd_up)
{
    double zup = (log10(m_wd_up) - B9O_MF_MU) / B9O_MF_SIGMA;
    double zlow = (-1.0 - B9O_MF_MU) / B9O_MF_SIGMA;
    double c = 1.0 / (B9O_MF_SIGMA * sqrt(2.0 * M_PI) * (Phi(zup) - Phi(zlow)));
    return log(c);
}

/* [RECALL] Cluster::logPriorMass: log-normal IMF in log10(m), expressed per unit mass. */
double b9o_log_prior_mass(double log_mass_norm, double m)
{
    double z = (log10(m) - B9O_MF_MU) / B9O_MF_SIGMA;
    return log_mass_norm - 0.5 * z * z - log(m) - log(B9O_LN10);
}

static double logaddexp(double a, double b)
{
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double hi = a > b ? a : b, lo = a > b ? b : a;
    return hi + log1p(exp(lo - hi));
}

/*
 * SURVEY 8a row a6 -- [RECALL] StellarSystem::logPost: mass prior + Gaussian terms over the
 * filters in use.  A non-finite predicted magnitude makes the star impossible (-inf).
 */
static double star_loglike(const b9_pack *p, const b9_stars *s, const b9o_iso *iso,
                           const double *par, double log_mass_norm, int i, double m1, double q)
{
    const int nf = p->n_filt;
    double pred[64];
    int wt = s->wd_type ? s->wd_type[i] : 0;
    combined_mags(p, iso, par, m1, q, wt, pred);
    double ll = b9o_log_prior_mass(log_mass_norm, m1);
    for (int f = 0; f < nf; ++f) {
        double sig = s->sigma[(size_t)i * nf + f];
        if (!isfinite(pred[f])) return -INFINITY;
        if (sig > 0.0) {
            double var = sig * sig, d = pred[f] - s->obs[(size_t)i * nf + f];
            ll -= 0.5 * (log(2.0 * M_PI * var) + d * d / var);
        }
    }
    return ll;
}

/* Cluster-level prior -- [RECALL] Cluster::logPrior.  Flat in logAge on its support,
 * Gaussian (unnormalised) on every parameter with a positive prior variance, A_V >= 0,
 * and, for two populations, 0 <= lambda <= 1. */
double b9o_log_prior_cluster(const b9_priors *pr, const double *par, int n_pops)
{
    if (!(par[B9_P_LOGAGE] >= pr->log_age_min && par[B9_P_LOGAGE] <= pr->log_age_max)) return -INFINITY;
    if (par[B9_P_ABS] < 0.0) return -INFINITY;
    if (n_pops == 2 && !(par[B9_P_LAMBDA] >= 0.0 && par[B9_P_LAMBDA] <= 1.0)) return -INFINITY;
    double lp = 0.0;
    for (int k = 0; k < B9_NPARAM; ++k) {
        if (k == B9_P_LOGAGE) continue;
        if (n_pops < 2 && (k == B9_P_Y2 || k == B9_P_LAMBDA)) continue;
        if (pr->var[k] > 0.0) {
            double d = par[k] - pr->mean[k];
            lp -= 0.5 * d * d / pr->var[k];
        }
    }
    return lp;
}

double b9o_log_field_like(const b9_stars *s)
{
    double l = 0.0;
    for (int f = 0; f < s->n_filt; ++f) l -= log(s->filter_prior_max[f] - s->filter_prior_min[f]);
    return l;
}

/* ---- marginalised mode ("marg.cpp-like") --------------------------------
 * Per star:  L_i = sum over primary-mass nodes M1 (iso_increm equal sub-steps inside every
 * EEP interval of the derived isochrone, left-endpoint rule, weight dM) of
 *     prior(M1) * (1/n_q) * sum over mass-ratio nodes q_j = j/n_q, j = 0..n_q-1, of
 *     prod_f N(obs_f | combined_f(M1, q_j M1), sigma_f^2)
 * A star of stage WD integrates M1 over (agb_tip, m_wd_up] in 8 * iso_increm equal steps with
 * no secondary.  DESIGN.md "Math / marginalised mode" is normative. */
static double star_marg_loglike(const b9_pack *p, const b9_stars *s, const b9o_iso *iso,
                                const double *par, double log_mass_norm, const b9_options *opt, int i)
{
    const int K = opt->marg_iso_increm > 0 ? opt->marg_iso_increm : 1;
    const int Q = opt->marg_n_q > 0 ? opt->marg_n_q : 1;
    double acc = -INFINITY;
    if (s->stage[i] == B9_STAGE_WD) {
        const int steps = 8 * K;
        double dM = (p->m_wd_up - iso->agb_tip) / steps;
        if (!(dM > 0.0)) return -INFINITY;
        for (int j = 1; j <= steps; ++j) {
            double m1 = iso->agb_tip + dM * j;
            double ll = star_loglike(p, s, iso, par, log_mass_norm, i, m1, 0.0);
            acc = logaddexp(acc, ll + log(dM));
        }
        return acc;
    }
    for (int e = 0; e + 1 < iso->n; ++e) {
        double d = iso->mass[e + 1] - iso->mass[e];
        if (!(d > 0.0)) continue;
        double dM = d / K;
        for (int k = 0; k < K; ++k) {
            double m1 = fma((double)k, dM, iso->mass[e]);
            for (int j = 0; j < Q; ++j) {
                double q = (double)j / (double)Q;
                double ll = star_loglike(p, s, iso, par, log_mass_norm, i, m1, q);
                acc = logaddexp(acc, ll + log(dM / Q));
            }
        }
    }
    return acc;
}

/*
 * SURVEY 8a rows a6, a8, a9 -- [RECALL] MpiMcmcApplication::logPostStep (+ the two-population
 * mixture of multiPopMcmc): per star  log( (1-p_i) fsLike + p_i L_i ),  summed over stars in
 * file order, plus the cluster prior.
 */
int b9o_logpost(const b9_pack *p, const b9_stars *s, const b9_priors *pr, const b9_options *opt,
                const double *params, int n_walkers, double *out_logpost, double *out_perstar)
{
    if (p->n_filt > 64 || p->n_filt != s->n_filt) return B9_ERR_INVALID;
    const int n_pops = opt->n_pops == 2 ? 2 : 1;
    const double lmn = b9o_log_mass_norm(p->m_wd_up);
    const double log_fs = b9o_log_field_like(s);

    for (int w = 0; w < n_walkers; ++w) {
        const double *par = params + (size_t)w * B9_NPARAM;
        double lp = b9o_log_prior_cluster(pr, par, n_pops);
        b9o_iso iso[2];
        int ok = isfinite(lp);
        int built = 0;
        for (int k = 0; ok && k < n_pops; ++k) {
            ok = b9o_derive_isochrone(p, par[B9_P_LOGAGE], par[B9_P_FEH],
                                      k ? par[B9_P_Y2] : par[B9_P_Y], &iso[k]);
            built = k + 1;
        }
        if (!ok) {
            out_logpost[w] = -INFINITY;
            if (out_perstar) for (int i = 0; i < s->n_stars; ++i) out_perstar[(size_t)w * s->n_stars + i] = -INFINITY;
            for (int k = 0; k < built; ++k) b9o_iso_free(&iso[k]);
            continue;
        }
        double total = 0.0;
        /* Timed "native" build only (-fopenmp): threads over stars, as the reference's thread pool does
         * [RECALL].  The checker build has no OpenMP: this pragma is then ignored and the sum is sequential. */
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : total) schedule(static)
#endif
        for (int i = 0; i < s->n_stars; ++i) {
            double ll[2];
            for (int k = 0; k < n_pops; ++k)
                ll[k] = (opt->mode == B9_MODE_MARGINALISED)
                      ? star_marg_loglike(p, s, &iso[k], par, lmn, opt, i)
                      : star_loglike(p, s, &iso[k], par, lmn, i, s->mass1[i], s->mass_ratio[i]);
            double l = ll[0];
            if (n_pops == 2)
                l = logaddexp(log(par[B9_P_LAMBDA]) + ll[0], log1p(-par[B9_P_LAMBDA]) + ll[1]);
            double pm = s->clust_prior[i];
            double v = logaddexp(log1p(-pm) + log_fs, log(pm) + l);
            if (out_perstar) out_perstar[(size_t)w * s->n_stars + i] = v;
            total += v;
        }
        out_logpost[w] = lp + total;
        for (int k = 0; k < n_pops; ++k) b9o_iso_free(&iso[k]);
    }
    return B9_OK;
}

/* ---- per-star mass draws (the sampleMass counterpart) -----------------------
 * Restates include/base9_hip.h :: b9_sample_mass: the Gumbel-max draw over the marginalisation grid,
 * nodes visited sequentially (the rule is an argmax, so the order is immaterial).  Philox4x32-10 as
 * published (Salmon et al. 2011); tests/test_mcmc.py pins the numpy twin to the Random123 vectors
 * and tests/test_oracle.py pins this one to the twin. */
static void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)c0 * 0xD2511F53ull, p1 = (uint64_t)c2 * 0xCD9E8D57ull;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void b9o_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

static double gumbel(uint64_t seed, uint64_t row, uint32_t star, uint64_t node, uint32_t pop)
{
    uint32_t r[4];
    philox4x32((uint32_t)row, star, (uint32_t)node, (uint32_t)(node >> 32) * 2u + pop,
               (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32) ^ (uint32_t)(row >> 32), r);
    uint64_t x = ((uint64_t)(r[0] >> 5) << 26) + (uint64_t)(r[1] >> 6);
    double u = ((double)x + 0.5) * (1.0 / 9007199254740992.0);
    return -log(-log(u));
}

/* out_margin (nullable): best key minus second-best key of every draw -- lets a test skip the draws
 * that another implementation's last-bit differences could legitimately flip. */
int b9o_sample_mass(const b9_pack *p, const b9_stars *s, const b9_options *opt, const double *params,
                    int n_rows, uint64_t seed, int64_t row0, double *out_mass, double *out_ratio,
                    double *out_member, int32_t *out_pop, double *out_margin)
{
    if (p->n_filt > 64 || p->n_filt != s->n_filt) return B9_ERR_INVALID;
    const int n_pops = opt->n_pops == 2 ? 2 : 1;
    const int K = opt->marg_iso_increm > 0 ? opt->marg_iso_increm : 1;
    const int Q = opt->marg_n_q > 0 ? opt->marg_n_q : 1;
    const double lmn = b9o_log_mass_norm(p->m_wd_up);
    const double log_fs = b9o_log_field_like(s);
    const int n = s->n_stars;
    for (int r = 0; r < n_rows; ++r) {
        const double *par = params + (size_t)r * B9_NPARAM;
        const uint64_t row = (uint64_t)(row0 + r);
        b9o_iso iso[2];
        int ok = 1, built = 0;
        for (int k = 0; ok && k < n_pops; ++k) {
            ok = b9o_derive_isochrone(p, par[B9_P_LOGAGE], par[B9_P_FEH], k ? par[B9_P_Y2] : par[B9_P_Y], &iso[k]);
            built = k + 1;
        }
        for (int i = 0; i < n; ++i) {
            const size_t o = (size_t)r * n + i;
            out_mass[o] = out_ratio[o] = out_member[o] = 0.0;
            if (out_pop) out_pop[o] = 0;
            if (out_margin) out_margin[o] = INFINITY;
            if (!ok) continue;
            double best = -INFINITY, second = -INFINITY, bm = 0.0, bq = 0.0, ll[2] = {-INFINITY, -INFINITY};
            int bp = 0;
            for (int k = 0; k < n_pops; ++k) {
                const double lw = n_pops == 2 ? (k ? log1p(-par[B9_P_LAMBDA]) : log(par[B9_P_LAMBDA])) : 0.0;
                double acc = -INFINITY;
#define B9O_NODE(term_, id_, m_, q_)                                                               \
    do {                                                                                           \
        double term = (term_);                                                                     \
        if (term > -INFINITY) {                                                                    \
            acc = logaddexp(acc, term);                                                            \
            double key = term + lw + gumbel(seed, row, (uint32_t)i, (uint64_t)(id_), (uint32_t)k); \
            if (key > best) { second = best; best = key; bm = (m_); bq = (q_); bp = k; }            \
            else if (key > second) second = key;                                                   \
        }                                                                                          \
    } while (0)
                if (s->stage[i] == B9_STAGE_WD) {
                    const int steps = 8 * K;
                    double dM = (p->m_wd_up - iso[k].agb_tip) / steps;
                    if (dM > 0.0)
                        for (int j = 1; j <= steps; ++j) {
                            double m1 = iso[k].agb_tip + dM * j;
                            B9O_NODE(star_loglike(p, s, &iso[k], par, lmn, i, m1, 0.0) + log(dM), j, m1, 0.0);
                        }
                } else {
                    for (int e = 0; e + 1 < iso[k].n; ++e) {
                        double d = iso[k].mass[e + 1] - iso[k].mass[e];
                        if (!(d > 0.0)) continue;
                        double dM = d / K;
                        for (int kk = 0; kk < K; ++kk) {
                            double m1 = fma((double)kk, dM, iso[k].mass[e]);
                            for (int j = 0; j < Q; ++j) {
                                double q = (double)j / (double)Q;
                                B9O_NODE(star_loglike(p, s, &iso[k], par, lmn, i, m1, q) + log(dM / Q),
                                         ((int64_t)e * K + kk) * Q + j, m1, q);
                            }
                        }
                    }
                }
#undef B9O_NODE
                ll[k] = acc;
            }
            double l = ll[0];
            if (n_pops == 2) l = logaddexp(log(par[B9_P_LAMBDA]) + ll[0], log1p(-par[B9_P_LAMBDA]) + ll[1]);
            double pm = s->clust_prior[i];
            double v = logaddexp(log1p(-pm) + log_fs, log(pm) + l);
            if (best > -INFINITY) { out_mass[o] = bm; out_ratio[o] = bq; if (out_pop) out_pop[o] = bp; }
            out_member[o] = (l > -INFINITY) ? exp(log(pm) + l - v) : 0.0;
            if (out_margin) out_margin[o] = best - second;
        }
        for (int k = 0; k < built; ++k) b9o_iso_free(&iso[k]);
    }
    return B9_OK;
}

/* Convenience for tests/makeCMD parity: derive into caller buffers. */
int b9o_derive_isochrone_flat(const b9_pack *p, const double *par, int pop, int cap,
                              double *out_mass, double *out_mags, int *out_first_eep, int *out_n,
                              double *out_agb_tip)
{
    b9o_iso iso;
    int ok = b9o_derive_isochrone(p, par[B9_P_LOGAGE], par[B9_P_FEH], pop ? par[B9_P_Y2] : par[B9_P_Y], &iso);
    *out_n = 0; *out_first_eep = 0; *out_agb_tip = 0.0;
    if (!ok) return B9_OK;
    if (iso.n > cap) { b9o_iso_free(&iso); return B9_ERR_CAPACITY; }
    memcpy(out_mass, iso.mass, sizeof(double) * (size_t)iso.n);
    memcpy(out_mags, iso.mags, sizeof(double) * (size_t)iso.n * p->n_filt);
    *out_n = iso.n; *out_first_eep = iso.first_eep; *out_agb_tip = iso.agb_tip;
    b9o_iso_free(&iso);
    return B9_OK;
}

/* Thread control for the timed build (no-ops without OpenMP). */
int b9o_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void b9o_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
