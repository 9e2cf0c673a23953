// b9_capi_stage.cpp -- b9_load_pack / b9_load_stars: validates the caller's host tables and stages them to HBM once
// (DESIGN.md "Data layout").  No CPU fallback exists: without a HIP device b9_ctx_create fails.
#include "b9_ctx.h"

using namespace b9i;

namespace {

template <class T>
int upload(b9_ctx *ctx, std::vector<void *> &owner, const T *src, size_t count, const T **out)
{
    void *d = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIPCHK(ctx, hipMalloc(&d, bytes));
    owner.push_back(d);
    if (count) HIPCHK(ctx, hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(d);
    return B9_OK;
}

bool ascending(const double *a, int n)
{
    for (int i = 1; i < n; ++i)
        if (!(a[i] > a[i - 1])) return false;
    return true;
}

int padded_filters(int nf) { return nf <= 4 ? 4 : (nf <= 8 ? 8 : 16); }

double Phi(double x) { return 0.5 * std::erfc(-x * M_SQRT1_2); }

// [RECALL] Cluster::setM_wd_up -- normalisation of the log-normal IMF on [0.1 Msun, M_wd_up]
double log_mass_norm(double m_wd_up)
{
    const double mu = -1.02, sg = 0.67729;
    double zup = (std::log10(m_wd_up) - mu) / sg, zlow = (-1.0 - mu) / sg;
    double c = 1.0 / (sg * std::sqrt(2.0 * M_PI) * (Phi(zup) - Phi(zlow)));
    return std::log(c);
}

double log_prior_mass(double lmn, double m)
{
    const double mu = -1.02, sg = 0.67729, ln10 = 2.302585092994045684;
    double z = (std::log10(m) - mu) / sg;
    return lmn - 0.5 * z * z - std::log(m) - std::log(ln10);
}

}  // namespace

namespace b9i {

// (Re)build the device star arrays.  Layout (DESIGN.md "Data layout"): singles and binaries are
// each sorted by primary mass and cut into 64-star chunks (one wave each, so a wave never mixes
// the two kinds and neighbouring lanes search neighbouring isochrone rows); binary chunks come
// first.  Unused slots of a partial chunk and the tail padding hold mass1 = +inf (skipped by the
// hot kernel) and perm = -1.
int build_stars(b9_ctx *ctx)
{
    const HostStars &h = ctx->hs;
    const int n = h.n, nf = h.nf, nfp = ctx->pk.nfp;
    if (nf != ctx->pk.nf) return fail(ctx, B9_ERR_INVALID, "stars and pack disagree on n_filt");
    free_all(ctx->star_allocs);
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        bool ba = h.q[a] > 0.0, bb = h.q[b] > 0.0;
        if (ba != bb) return !ba;
        return h.mass1[a] < h.mass1[b];
    });
    int n_single = 0;
    for (int i = 0; i < n; ++i) if (!(h.q[order[i]] > 0.0)) n_single = i + 1;
    const int cs = (n_single + 63) / 64, cb = (n - n_single + 63) / 64, ct = cs + cb;
    const int n_pad = std::max(256, (ct * 64 + 255) / 256 * 256);
    // slot -> star (or -1): all binary chunks, then all single chunks.  Workgroups are dispatched
    // in slot order, so the expensive binary waves start first and the kernel's tail consists of
    // cheap single-star waves.
    std::vector<int> slot(n_pad, -1);
    for (int cb_i = 0; cb_i < cb; ++cb_i)
        for (int j = 0; j < 64; ++j) { int k = n_single + cb_i * 64 + j; if (k < n) slot[cb_i * 64 + j] = order[k]; }
    for (int cs_i = 0; cs_i < cs; ++cs_i)
        for (int j = 0; j < 64; ++j) { int k = cs_i * 64 + j; if (k < n_single) slot[(cb + cs_i) * 64 + j] = order[k]; }
    (void)ct;
    double log_fs = 0.0;
    for (int f = 0; f < nf; ++f) log_fs -= std::log(h.fmax[f] - h.fmin[f]);

    std::vector<double> obs((size_t)nfp * n_pad, 0.0), w((size_t)nfp * n_pad, 0.0);
    std::vector<double> mass1(n_pad, INFINITY), q(n_pad, 0.0), c0(n_pad, 0.0), c0m(n_pad, 0.0), la(n_pad, -INFINITY), ea(n_pad, 0.0);
    std::vector<int> flags(n_pad, 0), permp(n_pad, -1);
    for (int i = 0; i < n_pad; ++i) {
        const int s = slot[i];
        if (s < 0) continue;
        double g = 0.0;
        for (int f = 0; f < nf; ++f) {
            double sig = h.sigma[(size_t)s * nf + f];
            // an unused filter (sigma <= 0) carries weight 0; its observation is stored as 0 so that whatever the
            // file holds there (99.999, NaN, ...) cannot turn 0 * d * d into NaN
            obs[B9_SIDX(nfp, f, i)] = sig > 0.0 ? h.obs[(size_t)s * nf + f] : 0.0;
            if (sig > 0.0) {
                double var = sig * sig;
                w[B9_SIDX(nfp, f, i)] = 1.0 / var;
                g -= 0.5 * std::log(2.0 * M_PI * var);
            }
        }
        mass1[i] = h.mass1[s];
        q[i] = h.q[s];
        const double pm = h.prior[s];
        c0m[i] = std::log(pm) + g;
        // (mass1 <= 0 is only accepted in the marginalised mode, which never reads c0)
        c0[i] = h.mass1[s] > 0.0 ? std::log(pm) + (log_prior_mass(ctx->pk.log_mass_norm, h.mass1[s]) + g) : -INFINITY;
        la[i] = std::log1p(-pm) + log_fs;
        ea[i] = std::exp(la[i]);
        flags[i] = (h.wd_type[s] > 0 ? 1 : 0) | (h.stage[s] << 8);
        permp[i] = s;
    }
    // slots in descending order of primary mass: the heavy-star workgroups of k_star_like take the leading run of stars
    // heavier than a walker's AGB tip (the WD / NS-BH branch) from this list
    std::vector<int> heavy_slot;
    heavy_slot.reserve(n);
    for (int i = 0; i < n_pad; ++i) if (slot[i] >= 0) heavy_slot.push_back(i);
    std::stable_sort(heavy_slot.begin(), heavy_slot.end(), [&](int a, int b) { return mass1[a] > mass1[b]; });
    std::vector<double> heavy_mass(std::max(n, 1), 0.0);
    for (int k = 0; k < n; ++k) heavy_mass[k] = mass1[heavy_slot[k]];

    // the heavy-order copy (DevStars::hv_*)
    const int hv_pad = std::max(64, (n + 63) / 64 * 64);
    std::vector<double> hv_obs((size_t)nfp * hv_pad, 0.0), hv_w((size_t)nfp * hv_pad, 0.0), hv_q(hv_pad, 0.0), hv_c0(hv_pad, 0.0), hv_la(hv_pad, -INFINITY);
    std::vector<int> hv_flags(hv_pad, 0), hv_perm(hv_pad, -1);
    for (int k = 0; k < n; ++k) {
        const int i = heavy_slot[k];
        for (int f = 0; f < nfp; ++f) { hv_obs[(size_t)f * hv_pad + k] = obs[B9_SIDX(nfp, f, i)]; hv_w[(size_t)f * hv_pad + k] = w[B9_SIDX(nfp, f, i)]; }
        hv_q[k] = q[i]; hv_c0[k] = c0[i]; hv_la[k] = la[i]; hv_flags[k] = flags[i]; hv_perm[k] = permp[i];
    }
    DevStars st{};
    st.n = n; st.n_pad = n_pad; st.hv_pad = hv_pad;
    int rc;
    if ((rc = upload(ctx, ctx->star_allocs, obs.data(), obs.size(), &st.obs))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, w.data(), w.size(), &st.w))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, mass1.data(), mass1.size(), &st.mass1))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, q.data(), q.size(), &st.q))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, c0.data(), c0.size(), &st.c0))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, c0m.data(), c0m.size(), &st.c0m))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, la.data(), la.size(), &st.la))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, ea.data(), ea.size(), &st.ea))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, flags.data(), flags.size(), &st.flags))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, permp.data(), permp.size(), &st.perm))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, heavy_mass.data(), heavy_mass.size(), &st.heavy_mass))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, heavy_slot.data(), heavy_slot.size(), &st.heavy_slot))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, hv_obs.data(), hv_obs.size(), &st.hv_obs))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, hv_w.data(), hv_w.size(), &st.hv_w))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, hv_q.data(), hv_q.size(), &st.hv_q))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, hv_c0.data(), hv_c0.size(), &st.hv_c0))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, hv_la.data(), hv_la.size(), &st.hv_la))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, hv_flags.data(), hv_flags.size(), &st.hv_flags))) return rc;
    if ((rc = upload(ctx, ctx->star_allocs, hv_perm.data(), hv_perm.size(), &st.hv_perm))) return rc;
    {
        std::vector<int> wd_slot;
        for (int i = 0; i < n_pad; ++i) if (slot[i] >= 0 && h.stage[slot[i]] == B9_STAGE_WD) wd_slot.push_back(i);
        st.n_wd = (int)wd_slot.size();
        if (wd_slot.empty()) wd_slot.push_back(0);
        if ((rc = upload(ctx, ctx->star_allocs, wd_slot.data(), wd_slot.size(), &st.wd_slot))) return rc;
    }
    {
        // The marginalised kernel's copy (DevStars::mg_*): stars other than WD-stage ones, sorted by the first principal
        // component of their magnitudes (an unused filter counts as its column's mean; power iteration on the covariance).
        std::vector<int> ms;
        for (int s = 0; s < n; ++s) if (h.stage[s] != B9_STAGE_WD) ms.push_back(s);
        const int n_ms = (int)ms.size();
        std::vector<double> mean(nf, 0.0), cnt(nf, 0.0);
        auto used = [&](int s, int f) { const double sg = h.sigma[(size_t)s * nf + f], o = h.obs[(size_t)s * nf + f]; return sg > 0.0 && std::isfinite(o); };
        for (int s : ms) for (int f = 0; f < nf; ++f) if (used(s, f)) { mean[f] += h.obs[(size_t)s * nf + f]; cnt[f] += 1.0; }
        for (int f = 0; f < nf; ++f) mean[f] = cnt[f] > 0.0 ? mean[f] / cnt[f] : 0.0;
        std::vector<double> cov((size_t)nf * nf, 0.0), x(nf);
        for (int s : ms) {
            for (int f = 0; f < nf; ++f) x[f] = used(s, f) ? h.obs[(size_t)s * nf + f] - mean[f] : 0.0;
            for (int a = 0; a < nf; ++a) for (int b = 0; b < nf; ++b) cov[(size_t)a * nf + b] += x[a] * x[b];
        }
        std::vector<double> pc(nf, 1.0), nx(nf);
        for (int it = 0; it < 200; ++it) {
            double nrm = 0.0;
            for (int a = 0; a < nf; ++a) { double t = 0.0; for (int b = 0; b < nf; ++b) t += cov[(size_t)a * nf + b] * pc[b]; nx[a] = t; nrm += t * t; }
            if (!(nrm > 0.0)) break;                                     // (all magnitudes equal: any order will do)
            nrm = std::sqrt(nrm);
            for (int a = 0; a < nf; ++a) pc[a] = nx[a] / nrm;
        }
        std::vector<double> key(n, 0.0);
        // key = the star's coefficient along that component, least squares over the filters it HAS (a missing filter must not
        // read as "average brightness": the star would land among strangers and widen their chunk's union)
        for (int s : ms) {
            double t = 0.0, nn = 0.0;
            for (int f = 0; f < nf; ++f) if (used(s, f)) { t += (h.obs[(size_t)s * nf + f] - mean[f]) * pc[f]; nn += pc[f] * pc[f]; }
            key[s] = nn > 0.0 ? t / nn : 0.0;
        }
        std::stable_sort(ms.begin(), ms.end(), [&](int a, int b) { return key[a] < key[b]; });
        const int n_mc = std::max(1, (n_ms + 63) / 64), mg_pad = n_mc * 64;
        std::vector<double> mg_obs((size_t)nfp * mg_pad, 0.0), mg_w((size_t)nfp * mg_pad, 0.0), mg_c0m(mg_pad, 0.0), mg_la(mg_pad, -INFINITY);
        std::vector<int> mg_perm(mg_pad, -1);
        std::vector<int> slot_of(n, -1);
        for (int i = 0; i < n_pad; ++i) if (slot[i] >= 0) slot_of[slot[i]] = i;
        for (int k = 0; k < n_ms; ++k) {
            const int i = slot_of[ms[k]];
            for (int f = 0; f < nfp; ++f) { mg_obs[B9_SIDX(nfp, f, k)] = obs[B9_SIDX(nfp, f, i)]; mg_w[B9_SIDX(nfp, f, k)] = w[B9_SIDX(nfp, f, i)]; }
            mg_c0m[k] = c0m[i]; mg_la[k] = la[i]; mg_perm[k] = ms[k];
        }
        // dispatch order of the chunks: descending photometric spread (10th to 90th percentile of the chunk's observations,
        // summed over the filters -- robust against the few field stars every chunk holds)
        std::vector<double> spread(n_mc, -1.0), v;
        for (int c = 0; c < n_mc; ++c) {
            double sp = 0.0;
            bool any = false;
            for (int f = 0; f < nf; ++f) {
                v.clear();
                for (int j = 0; j < 64; ++j) {
                    const int k = c * 64 + j;
                    if (mg_perm[k] >= 0 && mg_w[B9_SIDX(nfp, f, k)] > 0.0) v.push_back(mg_obs[B9_SIDX(nfp, f, k)]);
                }
                if (v.size() < 2) continue;
                std::sort(v.begin(), v.end());
                sp += v[(v.size() - 1) * 9 / 10] - v[(v.size() - 1) / 10];
                any = true;
            }
            if (any) spread[c] = sp;
        }
        std::vector<int> order(n_mc);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return spread[a] > spread[b]; });
        st.mg_pad = mg_pad;
        // what the kernel reads is the SCALED pair (sqrt(w), sqrt(w) obs): a term's chi^2 is then sum_f fma(sw, C_f, -so)^2 --
        // two instructions per filter instead of three (sqrt(w)^2 = w to 1 ulp; an unused filter has sw = so = 0)
        for (size_t k = 0; k < mg_w.size(); ++k) { const double sw = std::sqrt(mg_w[k]); mg_obs[k] = sw * mg_obs[k]; mg_w[k] = sw; }
        if ((rc = upload(ctx, ctx->star_allocs, mg_obs.data(), mg_obs.size(), &st.mg_so))) return rc;
        if ((rc = upload(ctx, ctx->star_allocs, mg_w.data(), mg_w.size(), &st.mg_sw))) return rc;
        if ((rc = upload(ctx, ctx->star_allocs, mg_c0m.data(), mg_c0m.size(), &st.mg_c0m))) return rc;
        if ((rc = upload(ctx, ctx->star_allocs, mg_la.data(), mg_la.size(), &st.mg_la))) return rc;
        if ((rc = upload(ctx, ctx->star_allocs, mg_perm.data(), mg_perm.size(), &st.mg_perm))) return rc;
        if ((rc = upload(ctx, ctx->star_allocs, order.data(), order.size(), &st.marg_order))) return rc;
    }
    st.mg_n_pieces = 0; st.mg_piece = nullptr; st.mg_share_base = nullptr;
    ctx->st = st;
    ctx->marg_order_spread = st.marg_order;
    ctx->marg_plan_ok = false;
    free_all(ctx->marg_plan_allocs);
    ctx->marg_cost.clear();
    ctx->n_wd_stage = 0;
    for (int i = 0; i < n; ++i) ctx->n_wd_stage += h.stage[i] == B9_STAGE_WD;
    ctx->stars_dirty = false;
    return B9_OK;
}

}  // namespace b9i

extern "C" {

int b9_load_pack(b9_ctx *ctx, const b9_pack *p)
{
    if (!ctx || !p) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (p->n_filt < 1 || p->n_filt > B9_MAX_FILT) return fail(ctx, B9_ERR_CAPACITY, "n_filt must be in [1, 16]");
    if (p->n_feh < 2 || p->n_age < 2 || p->n_y < 1) return fail(ctx, B9_ERR_INVALID, "grid needs >= 2 FeH and >= 2 ages");
    if (!p->feh || !p->log_age || !p->y || !p->iso_first_eep || !p->iso_n_eep || !p->iso_offset || !p->mass ||
        !p->mags || !p->abs_coeff)
        return fail(ctx, B9_ERR_INVALID, "NULL table pointer in pack");
    if (!ascending(p->feh, p->n_feh) || !ascending(p->log_age, p->n_age) || !ascending(p->y, p->n_y))
        return fail(ctx, B9_ERR_INVALID, "grid axes must be strictly ascending");
    const int n_iso = p->n_feh * p->n_y * p->n_age;
    int max_eep = 0;
    std::vector<double> tips(n_iso);
    for (int k = 0; k < n_iso; ++k) {
        const int n = p->iso_n_eep[k];
        const long long off = p->iso_offset[k];
        if (n < 2 || off < 0 || off + n > p->n_points) return fail(ctx, B9_ERR_INVALID, "isochrone index out of range");
        for (int e = 1; e < n; ++e)
            if (p->mass[off + e] < p->mass[off + e - 1]) return fail(ctx, B9_ERR_INVALID, "isochrone masses must not descend");
        max_eep = std::max(max_eep, n);
        tips[k] = p->mass[off + n - 1];
    }
    const bool has_wd = p->n_wc_mass >= 2 && p->n_at_teff >= 2 && p->n_at_logg >= 2 && p->n_at_type >= 1;
    const int n_tracks = has_wd ? std::max(1, p->n_wc_carb) * p->n_wc_mass : 0;
    std::vector<int> wc_n, wc_off;
    if (has_wd) {
        if (!p->wc_mass || !p->wc_n_age || !p->wc_offset || !p->wc_log_age || !p->wc_log_teff || !p->wc_log_radius || !p->at_logg ||
            !p->at_log_teff || !p->at_mags)
            return fail(ctx, B9_ERR_INVALID, "NULL WD table pointer in pack");
        if (!ascending(p->wc_mass, p->n_wc_mass) || !ascending(p->at_logg, p->n_at_logg) || !ascending(p->at_log_teff, p->n_at_teff) ||
            (p->n_wc_carb > 1 && !ascending(p->wc_carb, p->n_wc_carb)))
            return fail(ctx, B9_ERR_INVALID, "WD table axes must be strictly ascending");
        if (p->n_wc_points < 2 || p->n_wc_points > 0x7fffffffLL) return fail(ctx, B9_ERR_INVALID, "bad number of cooling-track points");
        for (int t = 0; t < n_tracks; ++t) {       // every (carbonicity, mass) node is a track with its own age axis
            const int n = p->wc_n_age[t];
            const long long off = p->wc_offset[t];
            if (n < 2 || off < 0 || off + n > p->n_wc_points) return fail(ctx, B9_ERR_INVALID, "cooling track index out of range");
            if (!ascending(p->wc_log_age + off, n)) return fail(ctx, B9_ERR_INVALID, "the cooling ages of a track must be strictly ascending");
            wc_n.push_back(n); wc_off.push_back((int)off);
        }
    }

    free_all(ctx->pack_allocs);
    ctx->have_pack = false;
    DevPack d{};
    d.nf = p->n_filt; d.nfp = padded_filters(p->n_filt);
    d.n_feh = p->n_feh; d.n_y = p->n_y; d.n_age = p->n_age; d.max_eep = max_eep;
    int rc;
    auto &A = ctx->pack_allocs;
    if ((rc = upload(ctx, A, p->feh, p->n_feh, &d.feh))) return rc;
    if ((rc = upload(ctx, A, p->y, p->n_y, &d.y))) return rc;
    if ((rc = upload(ctx, A, p->log_age, p->n_age, &d.log_age))) return rc;
    ctx->h_feh.assign(p->feh, p->feh + p->n_feh); ctx->h_y.assign(p->y, p->y + p->n_y); ctx->h_log_age.assign(p->log_age, p->log_age + p->n_age);
    if ((rc = upload(ctx, A, p->iso_first_eep, n_iso, &d.first))) return rc;
    if ((rc = upload(ctx, A, p->iso_n_eep, n_iso, &d.cnt))) return rc;
    std::vector<long long> off(p->iso_offset, p->iso_offset + n_iso);
    if ((rc = upload(ctx, A, off.data(), off.size(), &d.off))) return rc;
    if ((rc = upload(ctx, A, p->mass, (size_t)p->n_points, &d.mass))) return rc;
    {   // pad magnitude rows to nfp
        std::vector<double> mg((size_t)p->n_points * d.nfp, 0.0);
        for (long long i = 0; i < p->n_points; ++i)
            std::memcpy(&mg[(size_t)i * d.nfp], &p->mags[(size_t)i * d.nf], sizeof(double) * d.nf);
        if ((rc = upload(ctx, A, mg.data(), mg.size(), &d.mags))) return rc;
    }
    if ((rc = upload(ctx, A, tips.data(), tips.size(), &d.tips))) return rc;
    for (int f = 0; f < B9_MAX_FILT; ++f) d.abs_m1[f] = f < d.nf ? p->abs_coeff[f] - 1.0 : 0.0;
    if (has_wd) {
        d.n_wc_carb = std::max(1, p->n_wc_carb); d.n_wc_mass = p->n_wc_mass; d.n_wc_points = (int)p->n_wc_points;
        d.wc_n0 = wc_n[0]; d.wc_off0 = wc_off[0];
        d.wc_uniform = 1;                          // a rectangular table: every track repeats track 0's age axis
        for (int t = 1; t < n_tracks && d.wc_uniform; ++t)
            d.wc_uniform = wc_n[t] == wc_n[0] && std::memcmp(p->wc_log_age + wc_off[t], p->wc_log_age + wc_off[0], sizeof(double) * wc_n[0]) == 0;
        d.n_at_type = p->n_at_type; d.n_at_logg = p->n_at_logg; d.n_at_teff = p->n_at_teff;
        const double zero = 0.0;
        if ((rc = upload(ctx, A, p->n_wc_carb >= 1 ? p->wc_carb : &zero, (size_t)d.n_wc_carb, &d.wc_carb))) return rc;
        if ((rc = upload(ctx, A, p->wc_mass, p->n_wc_mass, &d.wc_mass))) return rc;
        {
            std::vector<double> packed(wc_n.size());
            for (size_t t = 0; t < wc_n.size(); ++t) {
                const unsigned long long w = (unsigned long long)(unsigned)wc_n[t] | ((unsigned long long)(unsigned)wc_off[t] << 32);
                std::memcpy(&packed[t], &w, sizeof w);
            }
            if ((rc = upload(ctx, A, packed.data(), packed.size(), &d.wc_track))) return rc;
        }
        if ((rc = upload(ctx, A, p->wc_log_age, (size_t)p->n_wc_points, &d.wc_log_age))) return rc;
        if ((rc = upload(ctx, A, p->wc_log_teff, (size_t)p->n_wc_points, &d.wc_log_teff))) return rc;
        if ((rc = upload(ctx, A, p->wc_log_radius, (size_t)p->n_wc_points, &d.wc_log_radius))) return rc;
        if ((rc = upload(ctx, A, p->at_logg, p->n_at_logg, &d.at_logg))) return rc;
        if ((rc = upload(ctx, A, p->at_log_teff, p->n_at_teff, &d.at_log_teff))) return rc;
        size_t nat = (size_t)d.n_at_type * d.n_at_logg * d.n_at_teff;
        std::vector<double> at(nat * d.nfp, 0.0);
        for (size_t i = 0; i < nat; ++i)
            std::memcpy(&at[i * d.nfp], &p->at_mags[i * d.nf], sizeof(double) * d.nf);
        if ((rc = upload(ctx, A, at.data(), at.size(), &d.at_mags))) return rc;
    }
    {   // the heavy-star role's LDS image of the axes (DevPack::heavy_const)
        std::vector<double> hc;
        auto seg = [&](int k, const double *src, size_t n) { d.hc_off[k] = (int)hc.size(); if (src && n) hc.insert(hc.end(), src, src + n); };
        seg(0, p->log_age, (size_t)p->n_age);
        d.hc_age_staged = has_wd && (d.wc_uniform || d.n_wc_points <= B9_WC_AGE_LDS_MAX) ? 1 : 0;
        if (d.hc_age_staged) seg(1, p->wc_log_age + (d.wc_uniform ? d.wc_off0 : 0), (size_t)(d.wc_uniform ? d.wc_n0 : d.n_wc_points));
        else seg(1, nullptr, 0);
        if (has_wd) {
            const double zero = 0.0;
            seg(2, p->wc_mass, (size_t)p->n_wc_mass);
            seg(3, p->n_wc_carb >= 1 ? p->wc_carb : &zero, (size_t)d.n_wc_carb);
            seg(4, p->at_log_teff, (size_t)p->n_at_teff);
            seg(5, p->at_logg, (size_t)p->n_at_logg);
            std::vector<double> packed(wc_n.size());
            for (size_t t = 0; t < wc_n.size(); ++t) {
                const unsigned long long w = (unsigned long long)(unsigned)wc_n[t] | ((unsigned long long)(unsigned)wc_off[t] << 32);
                std::memcpy(&packed[t], &w, sizeof w);
            }
            seg(6, packed.data(), packed.size());
        } else for (int k = 2; k < 7; ++k) seg(k, nullptr, 0);
        d.hc_len = (int)hc.size();
        if ((rc = upload(ctx, A, hc.data(), hc.size(), &d.heavy_const))) return rc;
    }
    d.ifmr_id = p->ifmr_id;
    d.m_wd_up = p->m_wd_up;
    d.log_mass_norm = log_mass_norm(p->m_wd_up);
    ctx->pk = d;
    ctx->have_pack = true;
    if (ctx->have_stars) ctx->stars_dirty = true;
    return B9_OK;
}

int b9_load_stars(b9_ctx *ctx, const b9_stars *s)
{
    if (!ctx || !s) return B9_ERR_INVALID;
    if (block_outstanding(ctx)) return fail(ctx, B9_ERR_STATE, kBlockOutstanding);
    if (s->n_stars < 1 || s->n_filt < 1 || s->n_filt > B9_MAX_FILT) return fail(ctx, B9_ERR_INVALID, "bad star or filter count");
    if (!s->obs || !s->sigma || !s->mass1 || !s->mass_ratio || !s->clust_prior || !s->filter_prior_min || !s->filter_prior_max)
        return fail(ctx, B9_ERR_INVALID, "NULL pointer in stars");
    HostStars &h = ctx->hs;
    const size_t n = s->n_stars, nf = s->n_filt;
    for (size_t f = 0; f < nf; ++f)
        if (!(s->filter_prior_max[f] > s->filter_prior_min[f])) return fail(ctx, B9_ERR_INVALID, "filter_prior_max must exceed filter_prior_min");
    for (size_t i = 0; i < n; ++i)
        if (!(s->clust_prior[i] > 0.0 && s->clust_prior[i] <= 1.0)) return fail(ctx, B9_ERR_INVALID, "clust_prior must be in (0, 1]");
    // a filter in use (sigma > 0) needs a finite observation and a sigma whose 1/sigma^2 is finite; NaN sigmas
    // are input errors, not "unused" (the .phot convention for unused is a negative sigma)
    for (size_t i = 0; i < n * nf; ++i) {
        const double sg = s->sigma[i];
        if (std::isnan(sg)) return fail(ctx, B9_ERR_INVALID, "sigma is NaN (use a negative sigma for an unused filter)");
        if (sg > 0.0 && (!(sg >= 1e-150) || std::isinf(sg) || !std::isfinite(s->obs[i])))
            return fail(ctx, B9_ERR_INVALID, "a filter in use needs a finite observation and 1e-150 <= sigma < inf");
    }
    // masses: a NaN mass1 would break the ordering the slot sort relies on, and a non-positive one has no mass prior;
    // the mass ratio is secondary / primary in [0, 1] (0 = single)
    // (whether mass1 must also be positive depends on the mode -- the marginalised mode only uses it as a hint --
    //  and is checked when the stars are staged: check_ready)
    for (size_t i = 0; i < n; ++i) {
        if (!std::isfinite(s->mass1[i])) return fail(ctx, B9_ERR_INVALID, "mass1 must be finite");
        if (!(s->mass_ratio[i] >= 0.0 && s->mass_ratio[i] <= 1.0)) return fail(ctx, B9_ERR_INVALID, "mass_ratio must be in [0, 1]");
    }
    h.n = (int)n; h.nf = (int)nf;
    h.obs.assign(s->obs, s->obs + n * nf);
    h.sigma.assign(s->sigma, s->sigma + n * nf);
    h.mass1.assign(s->mass1, s->mass1 + n);
    h.min_mass1 = *std::min_element(h.mass1.begin(), h.mass1.end());
    h.q.assign(s->mass_ratio, s->mass_ratio + n);
    h.prior.assign(s->clust_prior, s->clust_prior + n);
    h.fmin.assign(s->filter_prior_min, s->filter_prior_min + nf);
    h.fmax.assign(s->filter_prior_max, s->filter_prior_max + nf);
    if (s->stage) h.stage.assign(s->stage, s->stage + n); else h.stage.assign(n, B9_STAGE_MSRG);
    if (s->wd_type) h.wd_type.assign(s->wd_type, s->wd_type + n); else h.wd_type.assign(n, 0);
    ctx->have_stars = true;
    ctx->stars_dirty = true;
    return B9_OK;
}

}  // extern "C"
