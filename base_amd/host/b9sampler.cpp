// b9sampler.cpp -- see b9sampler.hpp.  Plain C++ above the C ABI; built with -ffp-contract=off so that summary_rows
// rounds like the device's block_summary_row.
#include "b9sampler.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace b9h {

namespace {

constexpr double kForget = 0.9;                    // per-block forgetting factor of the pooled adaptation moments
constexpr double kScaleMin = 1e-8, kScaleMax = 1e8;  // the global step scale stays finite whatever the acceptance does

[[noreturn]] void fail(const std::string &msg) { throw std::runtime_error(msg); }

bool cholesky(std::vector<double> &a, int d)      // in place, lower; false if not positive definite
{
    for (int j = 0; j < d; ++j) {
        double s = a[j * d + j];
        for (int k = 0; k < j; ++k) s -= a[j * d + k] * a[j * d + k];
        if (!(s > 0.0)) return false;
        a[j * d + j] = std::sqrt(s);
        for (int i = j + 1; i < d; ++i) {
            double t = a[i * d + j];
            for (int k = 0; k < j; ++k) t -= a[i * d + k] * a[j * d + k];
            a[i * d + j] = t / a[j * d + j];
        }
        for (int i = 0; i < j; ++i) a[i * d + j] = 0.0;
    }
    return true;
}

// ---- the GPU runner ------------------------------------------------------------------------------------------
class DeviceRunner final : public BlockRunner {
  public:
    DeviceRunner(b9_ctx *ctx, int n_local, const std::vector<int32_t> &ids, const std::vector<int32_t> &free_idx, uint64_t seed)
        : ctx_(ctx), W_(n_local), d_((int)free_idx.size()), ids_(ids), free_(free_idx), seed_(seed)
    {
        params_.assign((size_t)W_ * B9_NPARAM, 0.0);
        logpost_.assign(W_, 0.0);
    }

    void start(const double *params, const double *logpost) override
    {
        std::copy(params, params + (size_t)W_ * B9_NPARAM, params_.begin());
        std::copy(logpost, logpost + W_, logpost_.begin());
        first_ = true;
    }

    Submitted submit(int slot, const Job &job) override
    {
        Slot &s = slot_[slot & 1];
        const size_t n = (size_t)job.n_steps;
        s.params = params_; s.logpost = logpost_;           // inputs of the first block; outputs of every block
        s.chol.assign(job.chol, job.chol + (size_t)d_ * d_);
        s.origin.assign(job.origin, job.origin + d_);
        if (job.want_samples) { s.samples.resize(n * W_ * d_); s.lps.resize(n * W_); }
        s.rows.resize((size_t)W_ * B9_ROW_DOUBLES(d_));
        s.want_samples = job.want_samples;
        b9_mcmc_block &b = s.blk;
        b = b9_mcmc_block{};
        b.n_walkers = W_; b.n_free = d_; b.free_idx = free_.data(); b.chol = s.chol.data(); b.walker_ids = ids_.data();
        b.seed = seed_; b.step0 = job.step0; b.n_steps = job.n_steps;
        b.params = s.params.data(); b.logpost = s.logpost.data();
        b.samples = job.want_samples ? s.samples.data() : nullptr;
        b.lps = job.want_samples ? s.lps.data() : nullptr;
        // both evaluation modes: enqueued without waiting, continuing from the state the previous block leaves on the
        // device, summary rows condensed by the block's last launch (in HBM for an exchange that reads them there)
        b.flags = B9_BLOCK_ASYNC | (first_ ? 0 : B9_BLOCK_CONTINUE) | (job.want_device_rows ? B9_BLOCK_ROWS_EVENT : 0);
        b.row_origin = s.origin.data();
        b.rows = s.rows.data();
        const int rc = b9_mcmc_run_block(ctx_, &b);
        if (rc != B9_OK) fail(b9_last_error(ctx_));
        s.pending = true;
        first_ = false;
        Submitted r;
        if (job.want_device_rows) { r.d_rows = static_cast<const double *>(b.d_rows); r.rows_ready = b.rows_ready; }
        return r;
    }

    Done collect(int slot) override
    {
        Slot &s = slot_[slot & 1];
        if (s.pending) {
            if (b9_mcmc_wait(ctx_, &s.blk) != B9_OK) fail(b9_last_error(ctx_));
            s.pending = false;
            params_ = s.params; logpost_ = s.logpost;
        }
        Done dn;
        dn.params = s.params.data(); dn.logpost = s.logpost.data();
        const bool have_chain = s.blk.samples != nullptr;
        dn.samples = have_chain ? s.samples.data() : nullptr;
        dn.lps = have_chain ? s.lps.data() : nullptr;
        dn.rows = s.blk.rows ? s.rows.data() : nullptr;
        dn.n_accept = (long)s.blk.n_accept;
        return dn;
    }

    void logpost(const double *params, int n, double *out) override
    {
        if (b9_logpost(ctx_, params, n, out, nullptr) != B9_OK) fail(b9_last_error(ctx_));
    }

  private:
    struct Slot {
        std::vector<double> params, logpost, chol, origin, samples, lps, rows;
        b9_mcmc_block blk{};
        bool pending = false, want_samples = false;
    };
    b9_ctx *ctx_;
    int W_, d_;
    std::vector<int32_t> ids_, free_;
    uint64_t seed_;
    std::vector<double> params_, logpost_;
    Slot slot_[2];
    bool first_ = true;
};

// ---- the test seam: blocks and evaluations through caller-supplied callbacks -----------------------------------
class CallbackRunner final : public BlockRunner {
  public:
    CallbackRunner(b9h_block_fn run, b9h_logpost_fn eval, void *user, int n_local, const std::vector<int32_t> &ids,
                   const std::vector<int32_t> &free_idx, uint64_t seed)
        : run_(run), eval_(eval), user_(user), W_(n_local), d_((int)free_idx.size()), ids_(ids), free_(free_idx), seed_(seed)
    {
        params_.assign((size_t)W_ * B9_NPARAM, 0.0);
        logpost_.assign(W_, 0.0);
    }
    void start(const double *params, const double *logpost) override
    {
        std::copy(params, params + (size_t)W_ * B9_NPARAM, params_.begin());
        std::copy(logpost, logpost + W_, logpost_.begin());
    }
    Submitted submit(int slot, const Job &job) override
    {
        Slot &s = slot_[slot & 1];
        const size_t n = (size_t)job.n_steps;
        s.params.resize(params_.size()); s.logpost.resize(W_);
        s.samples.resize(n * W_ * d_); s.lps.resize(n * W_);
        int64_t acc = 0;
        if (run_(user_, params_.data(), logpost_.data(), ids_.data(), W_, free_.data(), d_, job.chol, seed_, job.step0, job.n_steps,
                 s.params.data(), s.logpost.data(), s.samples.data(), s.lps.data(), &acc) != 0)
            fail("block callback failed");
        s.n_accept = (long)acc;
        params_ = s.params; logpost_ = s.logpost;            // the callback is synchronous
        return Submitted{};
    }
    Done collect(int slot) override
    {
        Slot &s = slot_[slot & 1];
        return Done{s.params.data(), s.logpost.data(), s.samples.data(), s.lps.data(), nullptr, s.n_accept};
    }
    void logpost(const double *params, int n, double *out) override
    {
        if (eval_(user_, params, n, out) != 0) fail("log-posterior callback failed");
    }

  private:
    struct Slot { std::vector<double> params, logpost, samples, lps; long n_accept = 0; };
    b9h_block_fn run_;
    b9h_logpost_fn eval_;
    void *user_;
    int W_, d_;
    std::vector<int32_t> ids_, free_;
    uint64_t seed_;
    std::vector<double> params_, logpost_;
    Slot slot_[2];
};

class CallbackExchange final : public Exchange {
  public:
    CallbackExchange(b9h_gather_fn g, void *user, int rank, int world) : g_(g), user_(user), rank_(rank), world_(world) {}
    int rank() const override { return rank_; }
    int world() const override { return world_; }
    void start_host(int slot, const double *rows, size_t count) override
    {
        buf_[slot & 1].resize(count * world_);
        if (g_(user_, rows, count, buf_[slot & 1].data()) != 0) fail("all-gather callback failed");
    }
    const double *finish(int slot) override { return buf_[slot & 1].data(); }
    const char *name() const override { return "caller-supplied all-gather callback"; }

  private:
    b9h_gather_fn g_;
    void *user_;
    int rank_, world_;
    std::vector<double> buf_[2];
};

}  // namespace

std::unique_ptr<BlockRunner> make_device_runner(b9_ctx *ctx, int n_local, const std::vector<int32_t> &walker_ids,
                                                const std::vector<int32_t> &free_idx, uint64_t seed, int mode)
{
    (void)mode;      // (since ABI 3 both modes run pipelined blocks with device rows; kept in the signature for callers)
    return std::unique_ptr<BlockRunner>(new DeviceRunner(ctx, n_local, walker_ids, free_idx, seed));
}
std::unique_ptr<BlockRunner> make_callback_runner(b9h_block_fn run, b9h_logpost_fn eval, void *user, int n_local,
                                                  const std::vector<int32_t> &walker_ids, const std::vector<int32_t> &free_idx, uint64_t seed)
{
    return std::unique_ptr<BlockRunner>(new CallbackRunner(run, eval, user, n_local, walker_ids, free_idx, seed));
}
std::unique_ptr<Exchange> make_callback_exchange(b9h_gather_fn gather, void *user, int rank, int world)
{
    return std::unique_ptr<Exchange>(new CallbackExchange(gather, user, rank, world));
}

double step_scale_factor(double rate)
{
    if (rate < 0.02) return 0.2;
    if (rate < 0.10) return 0.5;
    if (rate < 0.20) return 0.8;
    if (rate > 0.90) return 4.0;
    if (rate > 0.70) return 2.0;
    if (rate > 0.50) return 1.5;
    if (rate > 0.35) return 1.2;
    return 1.0;
}

void summary_rows(const double *samples, const double *params_end, const double *logpost_end, int n_steps, int n_local, int d,
                  const double *origin, double *rows)
{
    const int len = B9_ROW_DOUBLES(d);
    std::vector<double> x((size_t)n_steps * d);
    for (int w = 0; w < n_local; ++w) {
        double *row = rows + (size_t)w * len;
        for (int s = 0; s < n_steps; ++s)
            for (int k = 0; k < d; ++k) x[(size_t)s * d + k] = samples[((size_t)s * n_local + w) * d + k] - origin[k];
        int moved = 0;
        for (int s = 1; s < n_steps; ++s) {
            bool diff = false;
            for (int k = 0; k < d; ++k) diff = diff || (x[(size_t)s * d + k] != x[(size_t)(s - 1) * d + k]);
            moved += diff ? 1 : 0;
        }
        row[0] = logpost_end[w];
        std::memcpy(row + 1, params_end + (size_t)w * B9_NPARAM, sizeof(double) * B9_NPARAM);
        row[13] = (double)moved;
        row[14] = (double)n_steps;
        for (int i = 0; i < d; ++i) {
            double acc = 0.0;
            for (int s = 0; s < n_steps; ++s) acc = acc + x[(size_t)s * d + i];
            row[15 + i] = acc;
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double acc = 0.0;
                for (int s = 0; s < n_steps; ++s) acc = acc + x[(size_t)s * d + i] * x[(size_t)s * d + j];
                row[15 + d + i * d + j] = acc;
            }
    }
}

// ---- the sampler ----------------------------------------------------------------------------------------------
WalkerSampler::WalkerSampler(const SamplerConfig &cfg, BlockRunner *runner, Exchange *exchange)
    : cfg_(cfg), runner_(runner), ex_(exchange)
{
    d_ = (int)cfg.free_idx.size();
    if (d_ < 1 || d_ > 11 || (int)cfg.step.size() != d_) fail("sampler: need 1..11 sampled parameters with one step size each");
    if (cfg.n_walkers < 1 || cfg.n_walkers % ex_->world()) fail("sampler: the number of walkers must be a multiple of the number of ranks");
    if (cfg.block < 1) fail("sampler: block must be positive");
    per_ = cfg.n_walkers / ex_->world();
    row_len_ = B9_ROW_DOUBLES(d_);
    ids_.resize(per_);
    for (int k = 0; k < per_; ++k) ids_[k] = ex_->rank() * per_ + k;
    chol_.assign((size_t)d_ * d_, 0.0);
    for (int i = 0; i < d_; ++i) chol_[i * d_ + i] = cfg.step[i];
    s1_.assign(d_, 0.0);
    s2_.assign((size_t)d_ * d_, 0.0);
    all_params_.assign((size_t)cfg.n_walkers * B9_NPARAM, 0.0);
    all_logpost_.assign(cfg.n_walkers, -INFINITY);
}

void WalkerSampler::initialise(const double *start)
{
    const int W = cfg_.n_walkers;
    // common origin of the pooled moments: the ensemble's starting mean (second moments about it do not cancel)
    origin_.assign(d_, 0.0);
    for (int i = 0; i < d_; ++i) {
        double s = 0.0;
        for (int w = 0; w < W; ++w) s += start[(size_t)w * B9_NPARAM + cfg_.free_idx[i]];
        origin_[i] = s / W;
    }
    std::vector<double> params((size_t)per_ * B9_NPARAM), lp(per_);
    for (int k = 0; k < per_; ++k)
        std::memcpy(&params[(size_t)k * B9_NPARAM], start + (size_t)ids_[k] * B9_NPARAM, sizeof(double) * B9_NPARAM);
    runner_->logpost(params.data(), per_, lp.data());
    runner_->start(params.data(), lp.data());
    // exchange [lp, position] so every rank knows every walker's starting state
    const int len = 1 + B9_NPARAM;
    std::vector<double> mine((size_t)per_ * len);
    for (int k = 0; k < per_; ++k) {
        mine[(size_t)k * len] = lp[k];
        std::memcpy(&mine[(size_t)k * len + 1], &params[(size_t)k * B9_NPARAM], sizeof(double) * B9_NPARAM);
    }
    ex_->start_host(0, mine.data(), mine.size());
    const double *all = ex_->finish(0);
    for (int w = 0; w < W; ++w) {
        all_logpost_[w] = all[(size_t)w * len];
        std::memcpy(&all_params_[(size_t)w * B9_NPARAM], all + (size_t)w * len + 1, sizeof(double) * B9_NPARAM);
    }
}

void WalkerSampler::run(long n_steps, bool adapt, const RecordFn &record)
{
    if (n_steps <= 0) return;
    std::vector<int> sizes;
    for (long left = n_steps; left > 0;) { sizes.push_back((int)std::min<long>(cfg_.block, left)); left -= sizes.back(); }
    const size_t B = sizes.size();
    std::vector<long> step0(B);
    { long s = step_; for (size_t b = 0; b < B; ++b) { step0[b] = s; s += sizes[b]; } }
    std::vector<double> chol_scaled((size_t)d_ * d_), host_rows;
    bool on_device[2] = {false, false};
    const size_t count = (size_t)per_ * row_len_;

    // B9_SAMPLER_TRACE=1: host time of each phase of every block, to stderr (diagnostic)
    const bool trace = std::getenv("B9_SAMPLER_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto enqueue = [&](size_t b) {
        const double t_a = trace ? now() : 0.0;
        struct Tr { bool on; double t; size_t b; decltype(now) &clk; ~Tr() { if (on) std::fprintf(stderr, "b9 sampler: block %zu enqueue %.1f us\n", b, clk() - t); } } tr{trace, t_a, b, now};
        for (size_t i = 0; i < chol_scaled.size(); ++i) chol_scaled[i] = scale_ * chol_[i];
        BlockRunner::Job job{step0[b], sizes[b], chol_scaled.data(), origin_.data(), (bool)record, ex_->reads_device_rows()};
        const BlockRunner::Submitted sub = runner_->submit((int)(b & 1), job);
        // rows that will be in HBM: the collective is enqueued NOW, stream-ordered behind the block, and runs beside
        // the next block's kernels without the host
        // (a one-rank exchange declines -- its rows arrive with the block's download -- unless it is an RCCL group)
        on_device[b & 1] = sub.d_rows && ex_->start_device((int)(b & 1), sub.d_rows, sub.rows_ready, count);
    };
    auto finish = [&](size_t b) {
        const double t_a = trace ? now() : 0.0;
        const BlockRunner::Done dn = runner_->collect((int)(b & 1));
        const double t_b = trace ? now() : 0.0;
        step_ += sizes[b];
        accepted_ += dn.n_accept;
        if (record) record(BlockRecord{step0[b], sizes[b], per_, d_, ids_.data(), dn.samples, dn.lps, adapt});
        if (!on_device[b & 1]) {
            const double *rows = dn.rows;
            if (!rows) {
                if (!dn.samples) fail("sampler: the runner returned neither summary rows nor the chain");
                host_rows.resize(count);
                summary_rows(dn.samples, dn.params, dn.logpost, sizes[b], per_, d_, origin_.data(), host_rows.data());
                rows = host_rows.data();
            }
            ex_->start_host((int)(b & 1), rows, count);
        }
        const double *all = ex_->finish((int)(b & 1));
        const double t_c = trace ? now() : 0.0;
        consume(all, sizes[b], adapt);
        if (trace) std::fprintf(stderr, "b9 sampler: block %zu collect %.1f us, record + exchange %.1f us, adapt %.1f us\n", b, t_b - t_a, t_c - t_b, now() - t_c);
    };

    // block b+1 is enqueued (continuing from block b's device-resident state) as soon as block b-1 has been collected
    // and its rows consumed, while block b is still running
    enqueue(0);
    for (size_t b = 0; b < B; ++b) {
        if (b >= 1) finish(b - 1);
        if (b + 1 < B) enqueue(b + 1);
    }
    finish(B - 1);
}

void WalkerSampler::consume(const double *rows, int n, bool adapt)
{
    const int W = cfg_.n_walkers;
    double moved = 0.0;
    for (int w = 0; w < W; ++w) {
        const double *r = rows + (size_t)w * row_len_;
        all_logpost_[w] = r[0];
        std::memcpy(&all_params_[(size_t)w * B9_NPARAM], r + 1, sizeof(double) * B9_NPARAM);
        moved += r[13];
    }
    if (!adapt) return;
    // pooled fraction of steps (after the block's first) on which a walker moved
    const double rate = moved / std::max(1.0, (double)W * (n - 1.0));
    if (n > 4) scale_ = std::min(std::max(scale_ * step_scale_factor(rate), kScaleMin), kScaleMax);
    adapt_shape(rows);
}

void WalkerSampler::adapt_shape(const double *rows)
{
    const int W = cfg_.n_walkers, d = d_;
    // exponentially forgotten sums (window ~ 1/(1-kForget) blocks): the start-up transient and the part of a degeneracy
    // ridge the ensemble has already left stop shaping the proposal.  Every rank pools the same rows in the same
    // (walker) order, so every rank derives the same factor.
    double n_new = 0.0;
    std::vector<double> a1(d, 0.0), a2((size_t)d * d, 0.0);
    for (int w = 0; w < W; ++w) {
        const double *r = rows + (size_t)w * row_len_;
        n_new += r[14];
        for (int i = 0; i < d; ++i) a1[i] += r[15 + i];
        for (int i = 0; i < d * d; ++i) a2[i] += r[15 + d + i];
    }
    n_mom_ = kForget * n_mom_ + n_new;
    for (int i = 0; i < d; ++i) s1_[i] = kForget * s1_[i] + a1[i];
    for (int i = 0; i < d * d; ++i) s2_[i] = kForget * s2_[i] + a2[i];
    if (!(n_mom_ > 20.0 * d)) return;
    std::vector<double> mean(d), cov((size_t)d * d);
    for (int i = 0; i < d; ++i) mean[i] = s1_[i] / n_mom_;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) cov[i * d + j] = (s2_[i * d + j] - n_mom_ * mean[i] * mean[j]) / (n_mom_ - 1.0) * (2.38 * 2.38 / d);
    // A chain that has hardly moved yet (bad starting scale) has a collapsed sample covariance: adopting it would freeze
    // the sampler.  Only take it once no direction is more than 100x narrower than the current (scaled) proposal is.
    for (int i = 0; i < d; ++i) {
        double cur = 0.0;
        for (int j = 0; j < d; ++j) cur += chol_[i * d + j] * chol_[i * d + j];
        if (!(cov[i * d + i] > 1e-4 * scale_ * scale_ * cur)) return;
    }
    for (int i = 0; i < d; ++i) cov[i * d + i] *= 1.0 + 1e-9;
    if (!cholesky(cov, d)) return;
    if (!shaped_) {
        // first switch from the diagonal start-up steps to a learnt shape: keep the volume of the scaled proposal (the
        // acceptance-tuned size carries over, only the shape changes)
        double lr = 0.0;
        for (int i = 0; i < d; ++i) lr += std::log(chol_[i * d + i]) - std::log(cov[i * d + i]);
        scale_ *= std::exp(lr / d);
        shaped_ = true;
    }
    chol_ = cov;                                   // `scale_` keeps multiplying it and keeps adapting
}

}  // namespace b9h
