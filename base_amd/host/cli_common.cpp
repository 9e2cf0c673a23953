#include "cli_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>

namespace b9h {

namespace {

// [RECALL] msRgbModel ids of base9.yaml; names are accepted as well
std::string ms_model_name(const std::string &v)
{
    static const char *by_id[] = {"girardi", "chaboyer", "yale", "dsed", "dsed", "parsec"};
    if (v.size() == 1 && v[0] >= '0' && v[0] <= '5') return by_id[v[0] - '0'];
    return v;
}
std::string wd_model_name(const std::string &v)
{
    static const char *by_id[] = {"wood", "montgomery", "althaus", "renedo"};
    if (v.size() == 1 && v[0] >= '0' && v[0] <= '3') return by_id[v[0] - '0'];
    return v;
}

}  // namespace

void open_session(Session &s, int argc, char **argv, int n_pops, bool need_phot)
{
    Settings &st = s.settings;
    st.parse_args(argc, argv);
    const std::string model_dir = st.str("general.files.modelDirectory");
    if (model_dir.empty()) throw std::runtime_error("no modelDirectory (use --modelDirectory or general.files.modelDirectory)");
    const std::string ms = ms_model_name(st.str("general.main_sequence.msRgbModel", "parsec"));
    const std::string wd = wd_model_name(st.str("general.white_dwarfs.wdModel", "montgomery"));
    s.output_base = st.str("general.files.outputFileBase", "base9");

    std::vector<std::string> filters;
    if (need_phot) {
        const std::string phot = st.str("general.files.photFile");
        if (phot.empty()) throw std::runtime_error("no photFile (use --photFile or general.files.photFile)");
        s.phot = read_photometry(phot, st.num("general.cluster.minMag", -1e300), st.num("general.cluster.maxMag", 1e300),
                                 (int)st.integer("general.cluster.index", 0));
        filters = s.phot.filters;
    } else {
        filters = model_filters(model_dir, ms);
    }
    s.pack = load_model_pack(model_dir, ms, wd, filters);
    s.pack.ifmr_id = (int)st.integer("general.white_dwarfs.ifmr", B9_IFMR_WILLIAMS);
    s.pack.m_wd_up = st.num("general.white_dwarfs.M_wd_up", 8.0);

    // ---- starting values and priors ([RECALL] general.cluster.starting / priors) -------------
    s.start.assign(B9_NPARAM, 0.0);
    struct P { int idx; const char *key; double def_start, def_sigma, step; };
    const P ps[] = {{B9_P_FEH, "Fe_H", 0.0, 0.3, 1e-3}, {B9_P_MOD, "distMod", 10.0, 0.3, 5e-4},
                    {B9_P_ABS, "Av", 0.1, 0.1, 3e-4}, {B9_P_Y, "Y", 0.27, 0.0, 3e-4},
                    {B9_P_CARBONICITY, "carbonicity", 0.38, 0.0, 1e-3}};
    s.start[B9_P_LOGAGE] = st.num("general.cluster.starting.logAge", 9.0);
    s.mcmc.free_idx = {B9_P_LOGAGE};
    s.mcmc.step = {st.num("singlePopMcmc.stepSize.logAge", 5e-4)};
    for (int k = 0; k < B9_NPARAM; ++k) { s.priors.mean[k] = 0.0; s.priors.var[k] = 0.0; }
    for (const P &p : ps) {
        const double mean = st.num(std::string("general.cluster.priors.means.") + p.key, p.def_start);
        const double sigma = st.num(std::string("general.cluster.priors.sigmas.") + p.key, p.def_sigma);
        s.start[p.idx] = st.num(std::string("general.cluster.starting.") + p.key, mean);
        s.priors.mean[p.idx] = mean;
        const bool on_grid = !(p.idx == B9_P_Y && s.pack.y.size() < 2) &&
                             !(p.idx == B9_P_CARBONICITY && s.pack.wc_carb.size() < 2);
        if (sigma > 0.0 && on_grid) {            // sigma <= 0: the parameter stays at its starting value
            s.priors.var[p.idx] = std::isfinite(sigma) ? sigma * sigma : 0.0;   // .inf -> flat prior
            s.mcmc.free_idx.push_back(p.idx);
            s.mcmc.step.push_back(st.num(std::string("singlePopMcmc.stepSize.") + p.key, p.step));
        }
    }
    s.start[B9_P_IFMR_INTERCEPT] = st.num("general.cluster.starting.ifmrIntercept", 0.77);
    s.start[B9_P_IFMR_SLOPE] = st.num("general.cluster.starting.ifmrSlope", 0.08);
    s.start[B9_P_IFMR_QUAD] = st.num("general.cluster.starting.ifmrQuadCoef", 0.0);
    if (n_pops == 2) {
        if (s.pack.y.size() < 2) throw std::runtime_error("multiPopMcmc needs a model pack with a helium axis");
        s.start[B9_P_Y] = st.num("multiPopMcmc.YA_start", s.pack.y.front() + 0.25 * (s.pack.y.back() - s.pack.y.front()));
        s.start[B9_P_Y2] = st.num("multiPopMcmc.YB_start", s.pack.y.front() + 0.75 * (s.pack.y.back() - s.pack.y.front()));
        s.start[B9_P_LAMBDA] = st.num("multiPopMcmc.lambda_start", 0.5);
        for (int idx : {B9_P_Y, B9_P_Y2, B9_P_LAMBDA}) {
            if (std::find(s.mcmc.free_idx.begin(), s.mcmc.free_idx.end(), idx) != s.mcmc.free_idx.end()) continue;
            s.mcmc.free_idx.push_back(idx);
            s.mcmc.step.push_back(idx == B9_P_LAMBDA ? st.num("multiPopMcmc.lambdaStep", 1e-3) : st.num("multiPopMcmc.YStep", 3e-4));
        }
    }
    s.priors.log_age_min = s.pack.log_age.front();
    s.priors.log_age_max = s.pack.log_age.back();
    s.mcmc.n_walkers = (int)st.integer("gpu.walkers", 1);
    s.mcmc.burn_iter = st.integer("singlePopMcmc.stage2IterMax", 2000);
    s.mcmc.run_iter = st.integer("singlePopMcmc.runIter", 10000);
    s.mcmc.thin = std::max<long>(1, st.integer("singlePopMcmc.thin", 1));
    s.mcmc.block = std::max<long>(1, st.integer("gpu.block", 50));
    s.mcmc.seed = (uint64_t)st.integer("general.seed", 73);
    s.mcmc.verbose = st.integer("general.verbose", 0) != 0;

    // ---- GPU context ----------------------------------------------------------------------------
    if (b9_ctx_create((int)st.integer("gpu.device", -1), &s.ctx) != B9_OK) throw std::runtime_error(b9_last_error(nullptr));
    auto check = [&](int rc) { if (rc != B9_OK) throw std::runtime_error(b9_last_error(s.ctx)); };
    b9_pack pv = s.pack.view();
    check(b9_load_pack(s.ctx, &pv));
    if (need_phot) {
        b9_stars sv = s.phot.view();
        check(b9_load_stars(s.ctx, &sv));
    }
    check(b9_set_priors(s.ctx, &s.priors));
    b9_options opt{B9_MODE_GIVEN_MASS, n_pops, 8, 8};
    check(b9_set_options(s.ctx, &opt));
}

int report_and_exit_code(const char *prog, const std::exception &e)
{
    std::fprintf(stderr, "%s: %s\n", prog, e.what());
    return 1;
}

}  // namespace b9h
