// b9sampler.hpp -- the walker-parallel adaptive Metropolis driver above the C ABI (SURVEY.md section 8 rows e and f-1;
// BASELINE.json north_star "Partition independent walkers/chains across the 8 GPUs of one node with an RCCL all-gather
// ... for the adaptive proposal step").  There is no reference counterpart for the walkers -- the reference runs one
// adaptive chain on CPU threads [RECALL]; its staged burn-in (adapt, then freeze) is kept: run(n, adapt = true) then
// run(n, adapt = false).
//
//   * W independent Metropolis chains ("walkers"); walker w lives on rank w / (W / world).
//   * Between adaptation points a walker needs nothing from any other: every rank advances its own walkers for `block`
//     steps with no communication (device-resident: b9_mcmc_run_block, one launch per step).
//   * Once per block every rank contributes ONE summary row per local walker -- [log-posterior, position, #moves, n,
//     sum x, sum x x^T], condensed on the GPU by the block's last launch -- to one all-gather (Exchange: RCCL over
//     xGMI reading the rows in HBM).  Every rank pools the rows in walker order and derives the same proposal factor
//     (adaptive Metropolis, 2.38^2/d, pooled moments with exponential forgetting) and the same global step scale.
//   * The exchange of block b is consumed after block b+1 has been enqueued, so its latency hides behind that block's
//     kernels: the proposal of block b+1 is adapted from the rows of blocks <= b-1 -- for every rank count, one rank
//     included.  Random numbers are counter-based (Philox4x32-10, key = seed, counter = (step, walker, draw)).  A
//     walker's chain is therefore the same bits whatever the number of ranks.
//   * run() drains the pipeline when it returns, so run(a); run(b) adapts at a different point than run(a + b) and gives
//     (equally valid) different chains.
#pragma once
#include "../../include/base9_hip.h"
#include "b9dist.hpp"

#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

namespace b9h {

struct SamplerConfig {
    int n_walkers = 8;                    // all ranks together; a multiple of the exchange's world
    std::vector<int32_t> free_idx;        // sampled parameters (B9_P_*), 1..11 of them
    std::vector<double> step;             // initial step size per sampled parameter
    uint64_t seed = 1234;
    int block = 50;                       // steps between adaptation points (= between exchanges)
};

// what a finished block hands to the caller's sink (local walkers only)
struct BlockRecord {
    long step0;                           // global number of the block's first step
    int n_steps, n_local, d;
    const int32_t *walker_ids;            // [n_local] global ids
    const double *samples;                // [n_steps][n_local][d]
    const double *lps;                    // [n_steps][n_local]
    bool adapting;
};
using RecordFn = std::function<void(const BlockRecord &)>;

// Advances the LOCAL walkers block by block with a fixed proposal factor.  Two implementations: the GPU
// (make_device_runner: b9_mcmc_run_block, device-resident, pipelined) and, for tests of THIS file's logic on machines
// without a GPU, caller-supplied callbacks (make_callback_runner) -- nothing in this library evaluates a likelihood on
// the CPU.
class BlockRunner {
  public:
    virtual ~BlockRunner() = default;
    struct Job {
        long step0;
        int n_steps;
        const double *chol;               // [d][d] scaled proposal factor
        const double *origin;             // [d] origin of the summary moments
        bool want_samples;                // the caller will read Done::samples / lps
        bool want_device_rows;            // an exchange will read the rows in HBM (Submitted::d_rows / rows_ready)
    };
    struct Submitted { const double *d_rows = nullptr; void *rows_ready = nullptr; };   // non-null: the rows will be in HBM
    struct Done {
        const double *params, *logpost;   // state of the local walkers after the block
        const double *samples, *lps;      // chain record (null unless want_samples or rows had to be condensed on the host)
        const double *rows;               // host copy of the summary rows, or null (then condense `samples`: summary_rows)
        long n_accept;
    };
    virtual void start(const double *params, const double *logpost) = 0;   // state before the first block
    virtual Submitted submit(int slot, const Job &job) = 0;                // at most two outstanding
    virtual Done collect(int slot) = 0;                                    // in submission order
    virtual void logpost(const double *params, int n, double *out) = 0;    // plain evaluation (initialise)
};

// mode = the B9_MODE_* the context's options select (given-mass: fused one-launch steps, pipelined blocks, device rows)
std::unique_ptr<BlockRunner> make_device_runner(b9_ctx *ctx, int n_local, const std::vector<int32_t> &walker_ids,
                                                const std::vector<int32_t> &free_idx, uint64_t seed, int mode);

// C callbacks of the test seam (see tests/test_sampler_host.py)
typedef int (*b9h_block_fn)(void *user, const double *params_in, const double *logpost_in, const int32_t *walker_ids, int n_local,
                            const int32_t *free_idx, int d, const double *chol, uint64_t seed, int64_t step0, int n_steps,
                            double *params_out, double *logpost_out, double *samples, double *lps, int64_t *n_accept);
typedef int (*b9h_logpost_fn)(void *user, const double *params, int n, double *out);
typedef int (*b9h_gather_fn)(void *user, const double *mine, size_t count, double *all);
std::unique_ptr<BlockRunner> make_callback_runner(b9h_block_fn run, b9h_logpost_fn eval, void *user, int n_local,
                                                  const std::vector<int32_t> &walker_ids, const std::vector<int32_t> &free_idx, uint64_t seed);
std::unique_ptr<Exchange> make_callback_exchange(b9h_gather_fn gather, void *user, int rank, int world);

// Host statement of the device's block summary (b9_mcmc_block::rows; same sums in the same order, so the same bits).
// samples: [n_steps][n_local][d]; params_end: [n_local][B9_NPARAM]; rows: [n_local][B9_ROW_DOUBLES(d)].
void summary_rows(const double *samples, const double *params_end, const double *logpost_end, int n_steps, int n_local, int d,
                  const double *origin, double *rows);

class WalkerSampler {
  public:
    WalkerSampler(const SamplerConfig &cfg, BlockRunner *runner, Exchange *exchange);
    // start: [n_walkers][B9_NPARAM], identical on every rank.  Evaluates the local walkers and exchanges the result.
    void initialise(const double *start);
    // n_steps in blocks of cfg.block, pipelined on the device; adapt = false freezes the proposal (the main run)
    void run(long n_steps, bool adapt, const RecordFn &record = nullptr);

    long steps() const { return step_; }
    long accepted_local() const { return accepted_; }          // accepted proposals of the LOCAL walkers
    double scale() const { return scale_; }
    const std::vector<double> &chol() const { return chol_; }
    const std::vector<double> &all_params() const { return all_params_; }       // [n_walkers][B9_NPARAM], as of the last exchange
    const std::vector<double> &all_logpost() const { return all_logpost_; }
    const std::vector<int32_t> &walker_ids() const { return ids_; }
    int n_local() const { return per_; }
    int d() const { return d_; }

  private:
    void consume(const double *rows, int n, bool adapt);
    void adapt_shape(const double *rows);

    SamplerConfig cfg_;
    BlockRunner *runner_;
    Exchange *ex_;
    int d_, per_, row_len_;
    std::vector<int32_t> ids_;
    std::vector<double> chol_, origin_, all_params_, all_logpost_;
    double scale_ = 1.0, n_mom_ = 0.0;
    std::vector<double> s1_, s2_;
    bool shaped_ = false;
    long step_ = 0, accepted_ = 0;
};

// multiplicative update of the global step scale from a block's pooled acceptance (target 0.2-0.35)
double step_scale_factor(double rate);

}  // namespace b9h
