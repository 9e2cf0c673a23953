// sampleMass -- per-star mass posterior ([RECALL] BASE-9 sampleMass): re-reads the cluster chain that
// singlePopMcmc wrote to <outputFileBase>.res and, for every main-run row and every star, draws the
// primary mass and the mass ratio from the star's conditional posterior and reports the membership
// probability.  All of it is one call per batch of rows to b9_sample_mass (the marginalised-mode
// kernel in its sampling form).  Writes
//   <outputFileBase>.massSamples   one line per chain row:  mass ratio  mass ratio ...  (star order of the .phot)
//   <outputFileBase>.membership    one line per chain row:  membership probability of every star
// Settings: sampleMass.margIsoIncrem (--margIsoIncrem, default 4 sub-steps per EEP interval),
//           sampleMass.nMassRatios (--nMassRatios, default 4), general.seed.
#include "cli_common.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>

int main(int argc, char **argv)
{
    try {
        b9h::Session s;
        b9h::open_session(s, argc, argv, 1, true);
        const int K = (int)s.settings.integer("sampleMass.margIsoIncrem", 4), Q = (int)s.settings.integer("sampleMass.nMassRatios", 4);
        if (K < 1 || Q < 1) throw std::runtime_error("margIsoIncrem and nMassRatios must be positive");
        b9_options opt{B9_MODE_GIVEN_MASS, 1, K, Q};
        if (b9_set_options(s.ctx, &opt) != B9_OK) throw std::runtime_error(b9_last_error(s.ctx));

        // ---- the chain: header names the sampled parameters, then "logPost stage"; stage 3 = main run
        const std::string res_path = s.output_base + ".res";
        std::ifstream in(res_path);
        if (!in) throw std::runtime_error("cannot read " + res_path + " (run singlePopMcmc first)");
        std::string line;
        do {                                                     // (leading "# ..." lines say how the chain was made)
            if (!std::getline(in, line)) throw std::runtime_error(res_path + " is empty");
        } while (!line.empty() && line[0] == '#');
        std::vector<int> col_param;
        {
            std::istringstream hs(line);
            std::string name;
            while (hs >> name) {
                if (name == "logPost" || name == "stage") { col_param.push_back(-1); continue; }
                int idx = -2;
                for (int k = 0; k < B9_NPARAM; ++k) if (name == b9h::param_name(k)) idx = k;
                if (idx < 0) throw std::runtime_error("unknown column '" + name + "' in " + res_path);
                col_param.push_back(idx);
            }
        }
        if (col_param.size() < 3) throw std::runtime_error(res_path + ": malformed header");
        std::vector<double> rows;
        while (std::getline(in, line)) {
            std::istringstream ls(line);
            std::vector<double> v(col_param.size());
            bool ok = true;
            for (double &x : v) ok = ok && (bool)(ls >> x);
            if (!ok) continue;
            if ((int)v.back() != 3) continue;
            std::vector<double> row(s.start.begin(), s.start.begin() + B9_NPARAM);
            for (size_t c = 0; c < col_param.size(); ++c) if (col_param[c] >= 0) row[col_param[c]] = v[c];
            rows.insert(rows.end(), row.begin(), row.end());
        }
        const long n_rows = (long)(rows.size() / B9_NPARAM);
        if (n_rows == 0) throw std::runtime_error(res_path + " holds no main-run (stage 3) rows");

        const int n = s.phot.n_stars();
        const std::string mp = s.output_base + ".massSamples", bp = s.output_base + ".membership";
        FILE *fm = std::fopen(mp.c_str(), "w"), *fb = std::fopen(bp.c_str(), "w");
        if (!fm || !fb) throw std::runtime_error("cannot write " + mp + " / " + bp);
        for (int i = 0; i < n; ++i) std::fprintf(fm, "%s%s_mass %s_massRatio", i ? " " : "", s.phot.ids[i].c_str(), s.phot.ids[i].c_str());
        std::fprintf(fm, "\n");
        for (int i = 0; i < n; ++i) std::fprintf(fb, "%s%s", i ? " " : "", s.phot.ids[i].c_str());
        std::fprintf(fb, "\n");
        const long batch = 256;
        std::vector<double> mass((size_t)batch * n), ratio((size_t)batch * n), member((size_t)batch * n);
        const auto t0 = std::chrono::steady_clock::now();
        for (long r0 = 0; r0 < n_rows; r0 += batch) {
            const long m = std::min(batch, n_rows - r0);
            if (b9_sample_mass(s.ctx, rows.data() + (size_t)r0 * B9_NPARAM, (int32_t)m, s.mcmc.seed, r0, mass.data(), ratio.data(),
                               member.data(), nullptr) != B9_OK)
                throw std::runtime_error(b9_last_error(s.ctx));
            for (long r = 0; r < m; ++r) {
                for (int i = 0; i < n; ++i)
                    std::fprintf(fm, "%s%.6f %.4f", i ? " " : "", mass[(size_t)r * n + i], ratio[(size_t)r * n + i]);
                std::fprintf(fm, "\n");
                for (int i = 0; i < n; ++i) std::fprintf(fb, "%s%.6f", i ? " " : "", member[(size_t)r * n + i]);
                std::fprintf(fb, "\n");
            }
        }
        std::fclose(fm); std::fclose(fb);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "sampleMass: %ld chain rows x %d stars (%d x %d mass / mass-ratio nodes per EEP interval) in %.3f s (%.3e star draws/s) -> %s, %s\n",
                     n_rows, n, K, Q, sec, (double)n_rows * n / sec, mp.c_str(), bp.c_str());
        return 0;
    } catch (const std::exception &e) {
        return b9h::report_and_exit_code("sampleMass", e);
    }
}
