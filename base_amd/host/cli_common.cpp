#include <signal.h>
#include "cli_common.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include <cerrno>

#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

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

    // ---- evaluation mode: every star at its catalogue (mass1, massRatio), or marginalised over both ---------------
    {
        std::string mode = st.str("gpu.mode", "givenMass");
        std::transform(mode.begin(), mode.end(), mode.begin(), ::tolower);
        if (st.integer("gpu.marginalise", 0) != 0 || mode == "marginalised" || mode == "marginalized" || mode == "marg")
            s.options.mode = B9_MODE_MARGINALISED;
        else if (mode != "givenmass" && mode != "given-mass" && mode != "given_mass")
            throw std::runtime_error("gpu.mode must be givenMass or marginalised");
        s.options.n_pops = n_pops;
        s.options.marg_iso_increm = (int32_t)std::max<long>(1, st.integer("sampleMass.margIsoIncrem", 8));
        s.options.marg_n_q = (int32_t)std::max<long>(1, st.integer("sampleMass.nMassRatios", 8));
    }
    // ---- GPU context (a rank of a --gpus N launch takes the GPU of its local rank) ------------------------------------
    rank_from_env(s.rank, s.world, s.local_rank);
    if (s.world > 1 && s.mcmc.n_walkers % s.world)
        throw std::runtime_error("the number of walkers (--walkers) must be a multiple of the number of GPUs (--gpus)");
    const int device = s.world > 1 ? s.local_rank : (int)st.integer("gpu.device", -1);
    if (b9_ctx_create(device, &s.ctx) != B9_OK) throw std::runtime_error(b9_last_error(nullptr));
    auto check = [&](int rc) { if (rc != B9_OK) throw std::runtime_error(b9_last_error(s.ctx)); };
    b9_pack pv = s.pack.view();
    check(b9_load_pack(s.ctx, &pv));
    if (need_phot) {
        b9_stars sv = s.phot.view();
        check(b9_load_stars(s.ctx, &sv));
    }
    check(b9_set_priors(s.ctx, &s.priors));
    check(b9_set_options(s.ctx, &s.options));
    // --tilesPerBlock n pins the size of the canonical tile groups a walker's star terms are summed in (b9_tuning).  A
    // walker's chain is the same bits for every --gpus either way -- the automatic size follows the catalogue, not the
    // number of walkers per GPU -- the option only chooses another (equally valid) rounding.  Only that field changes: the
    // B9_* environment overrides read at the context's creation stay in force.
    if (const long tpb = st.integer("gpu.tilesPerBlock", 0)) {
        b9_tuning t{};
        check(b9_get_tuning(s.ctx, &t));
        t.tiles_per_block = (int32_t)std::max<long>(0, tpb);
        check(b9_set_tuning(s.ctx, &t));
    }
}

namespace {

double env_seconds(const char *name, double def)
{
    const char *v = std::getenv(name);
    if (!v || !*v) return def;
    char *end = nullptr;
    const double x = std::strtod(v, &end);
    return (end && end != v && x >= 0.0) ? x : def;
}

void remove_bootstrap_dir(const std::string &dir, int n)
{
    // only what a launch of ours can have put there: the id file of THIS launch's nonce and the ranks' ready markers
    const char *nonce = std::getenv("B9_LAUNCH_NONCE");
    if (nonce) { std::remove((dir + "/rccl_id." + nonce).c_str()); std::remove((dir + "/rccl_id." + nonce + ".tmp").c_str()); }
    if (nonce) for (int r = 0; r < n; ++r) std::remove((dir + "/ready." + nonce + "." + std::to_string(r)).c_str());
    rmdir(dir.c_str());
}

// SIGINT / SIGTERM to the launcher: the wait loop below sees the flag, forwards SIGTERM to the ranks (SIGKILL 5 s later),
// reaps them and removes the bootstrap directory -- nothing is left parked in ncclCommInitRank
volatile sig_atomic_t g_launcher_signal = 0;
void on_launcher_signal(int sig) { g_launcher_signal = sig; }

}  // namespace

bool launch_ranks_if_requested(int argc, char **argv, int *exit_code)
{
    if (std::getenv("B9_RANK") || std::getenv("RANK")) return false;           // already a rank of some launcher
    Settings st;
    st.parse_args(argc, argv);                                                  // (flags and YAML only: nothing touches a GPU)
    int n = (int)st.integer("gpu.gpus", 1);
    // --forceRanks (or B9_FORCE_LAUNCHER=1): one GPU still takes the whole multi-rank route -- a child process started
    // before any GPU call, the id file, ncclCommInitRank (world 1), the device-row all-gather, the part-file merge --
    // so that route can be rehearsed (and is tested) on a one-GPU machine
    const char *fl = std::getenv("B9_FORCE_LAUNCHER");
    const bool force = st.integer("gpu.forceRanks", 0) != 0 || (fl && std::atoi(fl) != 0);
    if (n <= 1 && !force) return false;
    n = std::max(1, n);
    char dir[] = "/tmp/b9dist_XXXXXX";
    if (!mkdtemp(dir)) throw std::runtime_error("cannot create a directory for the RCCL bootstrap");
    const auto t0 = std::chrono::steady_clock::now();
    const std::string nonce = std::to_string((long)getpid()) + "_" + std::to_string((long long)t0.time_since_epoch().count());
    setenv("B9_LAUNCH_NONCE", nonce.c_str(), 1);
    // Deadlines.  B9_LAUNCH_TIMEOUT_S (default 300): every rank must have its communicator up (ready marker written after
    // ncclCommInitRank and the first collective) by then -- ncclCommInitRank itself never gives up on a peer that does
    // not come.  B9_RUN_TIMEOUT_S (default: none): the whole run.  Either one ends the launch: the ranks are killed and
    // the launcher exits 124.
    const double init_deadline = env_seconds("B9_LAUNCH_TIMEOUT_S", 300.0), run_deadline = env_seconds("B9_RUN_TIMEOUT_S", 0.0);
    struct sigaction sa{}, old_int{}, old_term{};
    sa.sa_handler = on_launcher_signal;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGINT, &sa, &old_int);
    sigaction(SIGTERM, &sa, &old_term);
    const pid_t launcher_pid = getpid();
    std::vector<pid_t> kids;
    for (int r = 0; r < n; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            // the ranks already started would wait for the missing ones: end them, reap them, leave nothing behind
            for (pid_t k : kids) kill(k, SIGTERM);
            for (pid_t k : kids) { int st_ = 0; waitpid(k, &st_, 0); }
            remove_bootstrap_dir(dir, n);
            throw std::runtime_error("fork failed");
        }
        if (pid == 0) {
            sigaction(SIGINT, &old_int, nullptr);
            sigaction(SIGTERM, &old_term, nullptr);
            prctl(PR_SET_PDEATHSIG, SIGTERM);                   // a launcher that dies without a word (SIGKILL) still ends its ranks
            if (getppid() != launcher_pid) _exit(143);          // (it died between fork and prctl)
            setenv("B9_RANK", std::to_string(r).c_str(), 1);
            setenv("B9_WORLD_SIZE", std::to_string(n).c_str(), 1);
            setenv("B9_LOCAL_RANK", std::to_string(r).c_str(), 1);
            setenv("B9_DIST_DIR", dir, 1);
            setenv("B9_LAUNCHER_OWNS_DIR", "1", 1);
            if (force) setenv("B9_FORCE_RANKS", "1", 1);
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);      // dmabuf IPC (what this driver supports) unless the user chose
            execv("/proc/self/exe", argv);
            std::perror("execv");
            _exit(127);
        }
        kids.push_back(pid);
    }
    // Wait for all of them, in whatever order they end.  A rank that fails takes the others with it (a peer blocked in
    // the communicator would wait for it for ever) -- exactly the processes forked above, by pid; the launch's exit code is
    // the FIRST failure seen (the peers then die of our SIGTERM: their 143 says nothing).
    int first_fail = 0;
    size_t left = kids.size();
    bool all_ready = false, killing = false;
    auto kill_time = t0;
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    auto end_all = [&](int sig) { for (pid_t k : kids) if (k > 0) kill(k, sig); };
    while (left > 0) {
        int status = 0;
        if (g_launcher_signal && !killing) {
            std::fprintf(stderr, "launcher: signal %d: ending the %zu rank(s)\n", (int)g_launcher_signal, left);
            if (!first_fail) first_fail = 128 + (int)g_launcher_signal;
            end_all(SIGTERM); killing = true; kill_time = std::chrono::steady_clock::now();
        }
        const pid_t k = waitpid(-1, &status, WNOHANG);
        if (k < 0) {
            if (errno == EINTR) continue;                       // a signal arrived: the ranks are still ours to reap
            if (!first_fail) first_fail = 1;                    // ECHILD: nothing left to wait for
            break;
        }
        if (k > 0) {
            auto it = std::find(kids.begin(), kids.end(), k);
            if (it == kids.end()) continue;
            *it = -1; --left;
            const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
            if (code != 0 && !first_fail) first_fail = code;
            if (code != 0 && !killing) { end_all(SIGTERM); killing = true; kill_time = std::chrono::steady_clock::now(); }
            continue;
        }
        if (!all_ready) {
            all_ready = true;
            for (int r = 0; r < n && all_ready; ++r) all_ready = access((std::string(dir) + "/ready." + nonce + "." + std::to_string(r)).c_str(), F_OK) == 0;
        }
        const bool late_start = !all_ready && init_deadline > 0.0 && elapsed() > init_deadline;
        const bool late_run = run_deadline > 0.0 && elapsed() > run_deadline;
        if ((late_start || late_run) && !killing) {
            std::fprintf(stderr, "launcher: %s after %.0f s (%s): ending the %zu remaining rank(s)\n",
                         late_start ? "not every rank brought its RCCL communicator up" : "the run did not finish", elapsed(),
                         late_start ? "B9_LAUNCH_TIMEOUT_S" : "B9_RUN_TIMEOUT_S", left);
            if (!first_fail) first_fail = 124;
            end_all(SIGTERM); killing = true; kill_time = std::chrono::steady_clock::now();
        }
        if (killing && std::chrono::duration<double>(std::chrono::steady_clock::now() - kill_time).count() > 5.0) end_all(SIGKILL);
        usleep(20000);
    }
    remove_bootstrap_dir(dir, n);
    sigaction(SIGINT, &old_int, nullptr);
    sigaction(SIGTERM, &old_term, nullptr);
    *exit_code = first_fail;
    return true;
}

void merge_result_parts(const std::string &final_path, int world, int per, long rows_per_part)
{
    // Interleave the parts in walker order.  Nothing is deleted unless the merged file was written completely: every
    // part must deliver exactly the rows its rank wrote (the same count on every rank), and every write is checked.
    std::vector<std::ifstream> parts;
    for (int r = 0; r < world; ++r) {
        parts.emplace_back(final_path + ".part" + std::to_string(r));
        if (!parts.back()) throw std::runtime_error("cannot read " + final_path + ".part" + std::to_string(r));
    }
    std::ofstream fin(final_path);
    if (!fin) throw std::runtime_error("cannot write " + final_path + " (the part files are kept)");
    std::string line;
    for (int r = 0; r < world; ++r)              // leading comment lines and the column header: part 0's
        while (std::getline(parts[r], line)) {
            if (r == 0) fin << line << "\n";
            if (line.empty() || line[0] != '#') break;
        }
    std::vector<long> got(world, 0);
    for (bool more = true; more;)
        for (int r = 0; r < world && more; ++r)
            for (int w = 0; w < per; ++w) {
                if (!std::getline(parts[r], line)) { more = false; break; }
                fin << line << "\n";
                ++got[r];
            }
    for (int r = 0; r < world; ++r) {
        while (std::getline(parts[r], line)) ++got[r];               // anything left over is a mismatch too
        if (got[r] != rows_per_part) {
            fin.close();
            std::remove(final_path.c_str());                          // no half-merged file is left behind
            throw std::runtime_error(final_path + ".part" + std::to_string(r) + " holds " + std::to_string(got[r]) + " rows, expected " +
                                     std::to_string(rows_per_part) + " (truncated or mismatched part; the part files are kept)");
        }
    }
    fin.flush();
    if (!fin) throw std::runtime_error("writing " + final_path + " failed (disk full?); the part files are kept");
    fin.close();
    parts.clear();
    for (int r = 0; r < world; ++r) std::remove((final_path + ".part" + std::to_string(r)).c_str());
}

McmcResult run_mcmc(Session &s, Exchange &ex, const std::vector<std::string> &columns)
{
    const McmcConfig &cfg = s.mcmc;
    const int W = cfg.n_walkers, d = (int)cfg.free_idx.size();
    SamplerConfig sc;
    sc.n_walkers = W; sc.free_idx = cfg.free_idx; sc.step = cfg.step; sc.seed = cfg.seed; sc.block = (int)cfg.block;
    const int per = W / ex.world();
    std::vector<int32_t> ids(per);
    for (int k = 0; k < per; ++k) ids[k] = ex.rank() * per + k;
    auto runner = make_device_runner(s.ctx, per, ids, cfg.free_idx, cfg.seed, s.options.mode);
    WalkerSampler sampler(sc, runner.get(), &ex);
    std::vector<double> start((size_t)W * B9_NPARAM);
    for (int w = 0; w < W; ++w) std::copy(s.start.begin(), s.start.begin() + B9_NPARAM, start.begin() + (size_t)w * B9_NPARAM);
    sampler.initialise(start.data());
    if (!std::isfinite(sampler.all_logpost()[0]))
        throw std::runtime_error("the starting parameters have zero posterior probability (outside the model grid or the prior support)");

    const std::string final_path = s.output_base + ".res";
    const bool parts_route = ex.world() > 1 || forced_ranks();      // (--forceRanks: one rank still writes a part and merges it)
    const std::string my_path = parts_route ? final_path + ".part" + std::to_string(ex.rank()) : final_path;
    // What the run sampled -- which posterior (the evaluation mode decides that), by which ABI -- goes to the sidecar
    // <base>.res.meta; the .res itself keeps the one-header-line layout [RECALL] that upstream's tools and
    // np.loadtxt(skiprows=1) expect.  --resComment puts the same text into the .res as a leading "# ..." line as well.
    std::string what = "base9_hip ABI " + std::to_string(b9_abi_version()) + "; mode=" +
                       (s.options.mode == B9_MODE_MARGINALISED
                            ? "marginalised (margIsoIncrem=" + std::to_string(s.options.marg_iso_increm) + ", nMassRatios=" + std::to_string(s.options.marg_n_q) + ")"
                            : std::string("givenMass")) +
                       "; populations=" + std::to_string(s.options.n_pops) + "; walkers=" + std::to_string(W);
    const bool in_file = s.settings.integer("gpu.resComment", 0) != 0;
    if (ex.rank() == 0) {
        std::ofstream meta(final_path + ".meta");
        meta << what << "\n";
        if (!meta) throw std::runtime_error("cannot write " + final_path + ".meta");
    }
    std::unique_ptr<ResultWriter> out(new ResultWriter(my_path, columns, in_file ? what : std::string()));
    long rows_written = 0;
    const long total = cfg.burn_iter + cfg.run_iter;
    std::vector<double> v(d);
    auto sink = [&](const BlockRecord &r) {
        for (int st = 0; st < r.n_steps; ++st) {
            const long step = r.step0 + st;
            if (step % cfg.thin) continue;
            const int stage = r.adapting ? (step < cfg.burn_iter / 2 ? 1 : 2) : 3;
            for (int w = 0; w < r.n_local; ++w) {
                std::copy(r.samples + ((size_t)st * r.n_local + w) * d, r.samples + ((size_t)st * r.n_local + w + 1) * d, v.begin());
                out->row(v, r.lps[(size_t)st * r.n_local + w], stage);
                ++rows_written;
            }
        }
        if (cfg.verbose && ex.rank() == 0)
            std::fprintf(stderr, "  step %ld/%ld  scale %.3g  logPost[0] %.4f\n", r.step0 + r.n_steps, total, sampler.scale(), sampler.all_logpost()[0]);
    };
    ex.barrier();
    const auto t0 = std::chrono::steady_clock::now();
    sampler.run(cfg.burn_iter, true, sink);          // the proposal adapts (pooled over all walkers of all ranks) ...
    sampler.run(cfg.run_iter, false, sink);          // ... and is frozen for the main run [RECALL]
    McmcResult res;
    res.seconds = ex.all_reduce_max(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    res.steps = sampler.steps();
    res.accepted = sampler.accepted_local();
    res.star_evals_per_s = (double)res.steps * W * s.phot.n_stars() / res.seconds;
    out.reset();
    if (parts_route) {
        ex.barrier();                                 // every part file is complete
        if (ex.rank() == 0) merge_result_parts(final_path, ex.world(), per, rows_written);
        ex.barrier();
    }
    return res;
}

int report_and_exit_code(const char *prog, const std::exception &e)
{
    std::fprintf(stderr, "%s: %s\n", prog, e.what());
    return 1;
}

}  // namespace b9h
