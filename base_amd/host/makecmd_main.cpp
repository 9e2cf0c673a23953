// makeCMD -- writes the isochrone (colour-magnitude diagram) of the starting cluster parameters
// to <outputFileBase>.cmd ([RECALL] BASE-9 makeCMD).  Uses b9_derive_isochrone: the same kernel
// the sampler uses every step.  Columns: EEP, mass, then one apparent magnitude per model filter
// (absolute magnitude + distance modulus + (A_f/A_V - 1) A_V).  MS/RGB only.
#include "cli_common.hpp"

#include <algorithm>
#include <stdexcept>
#include <cstdio>

int main(int argc, char **argv)
{
    try {
        b9h::Session s;
        b9h::open_session(s, argc, argv, 1, false);
        const int nf = (int)s.pack.filters.size();
        int cap = 0;
        for (int n : s.pack.iso_n_eep) cap = std::max(cap, n);
        std::vector<double> mass(cap), mags((size_t)cap * nf);
        int32_t first = 0, n = 0;
        double tip = 0.0;
        if (b9_derive_isochrone(s.ctx, s.start.data(), 0, cap, mass.data(), mags.data(), &first, &n, &tip) != B9_OK)
            throw std::runtime_error(b9_last_error(s.ctx));
        if (n == 0) throw std::runtime_error("the cluster parameters lie outside the model grid");
        const std::string path = s.output_base + ".cmd";
        FILE *f = std::fopen(path.c_str(), "w");
        if (!f) throw std::runtime_error("cannot write " + path);
        std::fprintf(f, "# logAge %.6f FeH %.6f Y %.6f modulus %.6f absorption %.6f agbTipMass %.10f\n", s.start[B9_P_LOGAGE],
                     s.start[B9_P_FEH], s.start[B9_P_Y], s.start[B9_P_MOD], s.start[B9_P_ABS], tip);
        std::fprintf(f, "%5s %14s", "EEP", "mass");
        for (auto &fl : s.pack.filters) std::fprintf(f, " %12s", fl.c_str());
        std::fprintf(f, "\n");
        for (int e = 0; e < n; ++e) {
            std::fprintf(f, "%5d %14.10f", first + e, mass[e]);
            for (int k = 0; k < nf; ++k)
                std::fprintf(f, " %12.8f", mags[(size_t)e * nf + k] + s.start[B9_P_MOD] + (s.pack.abs_coeff[k] - 1.0) * s.start[B9_P_ABS]);
            std::fprintf(f, "\n");
        }
        std::fclose(f);
        std::fprintf(stderr, "makeCMD: %d evolutionary points -> %s\n", n, path.c_str());
        return 0;
    } catch (const std::exception &e) {
        return b9h::report_and_exit_code("makeCMD", e);
    }
}
