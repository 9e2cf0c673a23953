// Sanitizer driver (CPU only): exercises the C++ host parsers and the C oracle under
// -fsanitize=address,undefined.  argv: <models dir> <ms model> <phot file> <yaml file>
// Prints "OK <n_stars> <logpost>" on success.  Never touches the GPU library's compute entry points.
#include "../../base_amd/host/b9host.hpp"

#include <cmath>
#include <cstdio>
#include <stdexcept>

extern "C" int b9o_logpost(const b9_pack *, const b9_stars *, const b9_priors *, const b9_options *, const double *, int,
                           double *, double *);

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    try {
        b9h::Settings st;
        char *sargv[] = {argv[0], (char *)"--config", argv[4], (char *)"--runIter", (char *)"7"};
        st.parse_args(5, sargv);
        if (st.integer("singlePopMcmc.runIter", 0) != 7) throw std::runtime_error("flag override failed");
        b9h::Photometry ph = b9h::read_photometry(argv[3]);
        b9h::ModelPack pk = b9h::load_model_pack(argv[1], argv[2], "montgomery", ph.filters);
        b9_pack pv = pk.view();
        b9_stars sv = ph.view();
        b9_priors pr{};
        pr.log_age_min = pk.log_age.front(); pr.log_age_max = pk.log_age.back();
        b9_options op{B9_MODE_GIVEN_MASS, 1, 2, 2};
        double row[B9_NPARAM] = {0};
        row[B9_P_LOGAGE] = st.num("general.cluster.starting.logAge", 9.0);
        row[B9_P_FEH] = st.num("general.cluster.starting.Fe_H", 0.0);
        row[B9_P_Y] = st.num("general.cluster.starting.Y", 0.27);
        row[B9_P_MOD] = st.num("general.cluster.starting.distMod", 10.0);
        row[B9_P_ABS] = st.num("general.cluster.starting.Av", 0.1);
        row[B9_P_CARBONICITY] = 0.38;
        double lp = 0.0;
        std::vector<double> per(ph.n_stars());
        if (b9o_logpost(&pv, &sv, &pr, &op, row, 1, &lp, per.data()) != 0) throw std::runtime_error("oracle failed");
        op.mode = B9_MODE_MARGINALISED;
        double lpm = 0.0;
        b9_stars few = sv; few.n_stars = ph.n_stars() < 5 ? ph.n_stars() : 5;
        if (b9o_logpost(&pv, &few, &pr, &op, row, 1, &lpm, nullptr) != 0) throw std::runtime_error("oracle (marg) failed");
        // malformed inputs must throw, not crash
        int threw = 0;
        try { b9h::load_model_pack(argv[1], argv[2], "montgomery", {"U", "NoSuchFilter"}); } catch (const std::exception &) { ++threw; }
        try { b9h::read_photometry(std::string(argv[3]) + ".missing"); } catch (const std::exception &) { ++threw; }
        try { b9h::Settings s2; char *a2[] = {argv[0], (char *)"--bogus", (char *)"1"}; s2.parse_args(3, a2); } catch (const std::exception &) { ++threw; }
        if (threw != 3) throw std::runtime_error("error paths did not throw");
        if (!std::isfinite(lp) || !std::isfinite(lpm)) throw std::runtime_error("non-finite log-posterior");
        std::printf("OK %d %.6f %.6f\n", ph.n_stars(), lp, lpm);
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "driver: %s\n", e.what());
        return 1;
    }
}
