// cli_common.hpp -- what singlePopMcmc / multiPopMcmc / makeCMD / sampleMass share: settings -> context, and the
// sampling run (walker-parallel over the GPUs of one node when launched with --gpus N).
#pragma once
#include "b9host.hpp"
#include "b9sampler.hpp"

namespace b9h {

struct McmcConfig {
    std::vector<int32_t> free_idx;        // sampled parameters (B9_P_*)
    std::vector<double> step;             // initial step size per sampled parameter
    int n_walkers = 1;                    // all ranks together
    long burn_iter = 2000, run_iter = 10000, thin = 1, block = 50;
    uint64_t seed = 73;
    bool verbose = false;
};

struct McmcResult {
    long accepted = 0, steps = 0;         // accepted: this rank's walkers
    double seconds = 0.0, star_evals_per_s = 0.0;   // whole job (all ranks), max-over-ranks time
};

struct Session {
    Settings settings;
    ModelPack pack;
    Photometry phot;
    b9_ctx *ctx = nullptr;
    std::vector<double> start;        // B9_NPARAM starting row
    b9_priors priors{};
    b9_options options{B9_MODE_GIVEN_MASS, 1, 8, 8};
    McmcConfig mcmc;
    std::string output_base;
    int rank = 0, world = 1, local_rank = 0;      // this process within a --gpus N launch
    ~Session() { if (ctx) b9_ctx_destroy(ctx); }
};

// Parses flags / YAML, loads the photometry (unless `need_phot` is false) and the model pack with
// the photometry's filters, creates the GPU context and stages everything.  Throws on error.
void open_session(Session &s, int argc, char **argv, int n_pops, bool need_phot);

// --gpus N (or gpu.gpus): when this process is not yet a rank of a launch, starts N copies of itself -- one per GPU, with
// B9_RANK / B9_WORLD_SIZE / B9_LOCAL_RANK / B9_DIST_DIR set -- BEFORE anything touches a GPU, waits for them and
// returns their worst exit code through *exit_code (true = the caller is the launcher and must exit with that code).
bool launch_ranks_if_requested(int argc, char **argv, int *exit_code);

// Adaptive Metropolis ([RECALL] the staged burn-in of MpiMcmcApplication): burn-in with the proposal adapted after
// every block from the pooled rows of all walkers (all GPUs), then the main run with the proposal frozen.  Burn-in rows
// are written with stage 1..2, the main run with stage 3, walkers in id order within a step; with several ranks every
// rank writes a part file and rank 0 merges them into <output_base>.res.
McmcResult run_mcmc(Session &s, Exchange &exchange, const std::vector<std::string> &columns);

// Rank 0's merge of <final_path>.part<r>, r < world (each `rows_per_part` data rows, `per` walkers per step and rank), into
// <final_path>: comment + header of part 0, then the steps' rows in walker order.  Throws -- keeping every part file and
// leaving no merged file -- when a part is short, long or unreadable, or the output cannot be written; removes the parts
// only after the merged file is complete.
void merge_result_parts(const std::string &final_path, int world, int per, long rows_per_part);

int report_and_exit_code(const char *prog, const std::exception &e);

}  // namespace b9h
