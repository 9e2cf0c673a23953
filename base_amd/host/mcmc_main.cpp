// singlePopMcmc / multiPopMcmc -- compiled twice (-DB9_N_POPS=1 / 2).  Command-line surface and
// files follow BASE-9 [RECALL]: --config base9.yaml plus long flags that override it; reads the
// .phot, loads the model pack, samples, writes <outputFileBase>.res.
//
// Additions of this build (no reference counterpart): --walkers W independent chains with a pooled adaptive
// proposal, --gpus N to shard them over the GPUs of one node (one process per GPU, started by this program itself;
// the per-block all-gather of the walkers' summary rows is RCCL over xGMI), --marginalise / --mode marginalised to
// integrate every star over primary mass and mass ratio instead of conditioning on the catalogue's masses.
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
        int code = 0;
        if (b9h::launch_ranks_if_requested(argc, argv, &code)) return code;     // the launcher of a --gpus N run
        {
            int r0 = 0, w0 = 1, l0 = 0;
            b9h::rank_from_env(r0, w0, l0);
            b9h::test_stall("start", r0);                                        // (test hook; nothing has touched a GPU yet)
        }
        b9h::Session s;
        b9h::open_session(s, argc, argv, B9_N_POPS, true);
        std::unique_ptr<b9h::Exchange> ex = (s.world > 1 || b9h::forced_ranks()) ? b9h::make_rccl_exchange(s.rank, s.world, b9h::default_bootstrap_dir(), b9_device_id(s.ctx))
                                                         : b9h::make_local_exchange();
        std::vector<std::string> cols;
        for (int idx : s.mcmc.free_idx) cols.push_back(B9_N_POPS == 2 && idx == B9_P_Y ? "YA" : b9h::param_name(idx));
        if (s.rank == 0) {
            std::fprintf(stderr, "%s: %d stars x %zu filters, %d walker(s) on %d GPU(s) [%s", prog, s.phot.n_stars(),
                         s.phot.filters.size(), s.mcmc.n_walkers, s.world, ex->name());
            if (ex->comm_ranks() > 0) std::fprintf(stderr, "; communicator of %d rank(s) on %s", ex->comm_ranks(), ex->devices().c_str());
            std::fprintf(stderr, "], %s mode, sampling", s.options.mode == B9_MODE_MARGINALISED ? "marginalised" : "given-mass");
            for (auto &c : cols) std::fprintf(stderr, " %s", c.c_str());
            std::fprintf(stderr, "\n");
        }
        b9h::McmcResult r = b9h::run_mcmc(s, *ex, cols);
        if (s.rank == 0)
            std::fprintf(stderr, "%s: %ld steps in %.3f s (%.0f steps/s, %.3e star-likelihood evals/s), acceptance %.3f -> %s.res\n",
                         prog, r.steps, r.seconds, r.steps / r.seconds, r.star_evals_per_s,
                         (double)r.accepted / ((double)r.steps * (s.mcmc.n_walkers / s.world)), s.output_base.c_str());
        return 0;
    } catch (const std::exception &e) {
        return b9h::report_and_exit_code(prog, e);
    }
}
