// singlePopMcmc / multiPopMcmc -- compiled twice (-DB9_N_POPS=1 / 2).  Command-line surface and
// files follow BASE-9 [RECALL]: --config base9.yaml plus long flags that override it; reads the
// .phot, loads the model pack, samples, writes <outputFileBase>.res.
#include "cli_common.hpp"

#include <algorithm>
#include <cstdio>

#ifndef B9_N_POPS
#define B9_N_POPS 1
#endif

int main(int argc, char **argv)
{
    const char *prog = B9_N_POPS == 2 ? "multiPopMcmc" : "singlePopMcmc";
    try {
        b9h::Session s;
        b9h::open_session(s, argc, argv, B9_N_POPS, true);
        std::vector<std::string> cols;
        for (int idx : s.mcmc.free_idx) cols.push_back(B9_N_POPS == 2 && idx == B9_P_Y ? "YA" : b9h::param_name(idx));
        b9h::ResultWriter out(s.output_base + ".res", cols);
        std::fprintf(stderr, "%s: %d stars x %zu filters, %d walker(s), sampling", prog, s.phot.n_stars(), s.phot.filters.size(), s.mcmc.n_walkers);
        for (auto &c : cols) std::fprintf(stderr, " %s", c.c_str());
        std::fprintf(stderr, "\n");
        b9h::McmcResult r = b9h::run_mcmc(s.ctx, s.mcmc, s.start, s.phot.n_stars(), &out);
        std::fprintf(stderr, "%s: %ld steps in %.3f s (%.0f steps/s, %.3e star-likelihood evals/s), acceptance %.3f -> %s.res\n",
                     prog, r.steps, r.seconds, r.steps / r.seconds, r.star_evals_per_s,
                     (double)r.accepted / ((double)r.steps * s.mcmc.n_walkers), s.output_base.c_str());
        return 0;
    } catch (const std::exception &e) {
        return b9h::report_and_exit_code(prog, e);
    }
}
