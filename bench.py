#!/usr/bin/env python3
"""bench.py -- star-likelihood evals/sec of the MI355X-native BASE-9 log-posterior path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric: "star-likelihood evals/sec ... on synthetic 50k-star x 8-filter clusters at 1 GPU,
with 1/2/4/8-GPU walker-parallel throughput"; configs[2] sharded 8 ways): 50 000 stars x 8 filters, PARSEC-shaped
synthetic pack, 8 walkers per GPU (weak scaling: 64 walkers at 8 GPUs).  One "step" is one adaptive-Metropolis step of
every walker: propose -> log-posterior of the rank's walkers -> accept/reject, all on the GPU (ONE HIP launch per step
behind the C ABI, b9_mcmc_run_block); every 100 steps the ranks exchange one all-gather of per-walker summary rows --
condensed on the GPU, gathered from HBM by RCCL over xGMI -- and re-derive the pooled proposal covariance.  The driver
is the C++ host library (base_amd/host/b9sampler.cpp, b9dist.cpp): no torch, no Python in the timed loop beyond one
ctypes call.  Star data and model tables are resident in HBM before the timed region starts.

Launching.  `python bench.py --gpus N ...` invoked plainly starts its own N rank processes (one per GPU) before
anything touches a GPU, waits for them (start-up deadline B9_LAUNCH_TIMEOUT_S, default 300 s, for every rank's RCCL
communicator to come up; optional whole-run deadline B9_RUN_TIMEOUT_S) and relays rank 0's JSON line.  Under a launcher
that already exports RANK / WORLD_SIZE / LOCAL_RANK (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N
...) the process is a rank.  `--gpus 1 --force-ranks` sends ONE rank down the whole multi-rank route (child process
started before any GPU call, id file, ncclCommInitRank with world 1, RCCL all-gather of the device rows) -- the
rehearsal of that route on a one-GPU box.

value        = n_stars x total walkers x K / max-over-ranks wall time   (whole job, all GPUs)
roofline     = the dominant kernel (k_mcmc_step, the fused sampler step).  The kernel is NOT HBM-bound: the walkers of
               a GPU share every star tile through the XCD-local L2, and its waves spend their time in dependent
               fp64 VALU chains.  Reported, each against its own peak: fp64 VALU issue (the bound the counters name;
               `frac`), the USEFUL part of it (`useful_frac`: the algorithmic fp64 operations of DESIGN.md section 3
               over the same peak), HBM traffic (counters), and the algorithmic byte rate of SURVEY.md 8(d).
               Launch duration is measured live with HIP events on the launch stream; instruction and byte counts per
               launch come from the committed rocprofv3 PMC passes of this same command (profiles/, tagged) and are
               dropped (null, with the reason) when the kernel sources have changed since they were measured.
cpu_baseline = the CPU oracle ("port"; the reference itself is not mounted) timed on this
               box's host cores on a bounded sample of the same workload (rank 0, N=1); with it, |delta logPost| of
               the HIP path against that CPU path: of b9_logpost on 128 random rows AND of the timed sampler's own
               ensemble state after the timed region (delta_logpost_sampler).
sustained    = the same loop over 140 000 further steps (>= 2 s), timed the same way (its own value / ms_per_step; `value` stays the
               K steps'), with the shader clock observed in-kernel over that stretch (`clock_mhz_observed`): the roofline
               peaks are quoted at the 2.4 GHz maximum, `roofline.*_at_observed_clock` rescale them.
groups       = with N > 1 ranks the line is refused (exit 3) unless the RCCL communicator itself reports N ranks on N
               DISTINCT GPUs (PCI bus ids gathered through it).
marginalised_mode = the same objects (value, roofline, cpu_baseline + delta, the sampler with its own sustained leg and
               clock) for the marginalised evaluation mode.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
CLOCK_GHZ, N_SIMD = 2.4, 1024   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz max clock; an fp64 VALU wave-instruction issues in 4 cycles
N_STARS, N_FILT, WALKERS_PER_GPU = 50000, 8, 8
MCMC_BLOCK = 100        # steps between adaptation points (= between all-gathers)
TIMING_EVERY = 25       # a HIP-event bracket opens at every 25th launch of the dominant kernel in the timed region and spans 8 launches
PREWARM_STEPS = 1000    # untimed, BEFORE the W warm-up steps: clocks, first touch of every buffer, RCCL channels, and the sampler's
                        # own burn-in (10 adaptation blocks: the timed steps run with the adapted proposal, as a real run's do)
SUSTAINED_STEPS = 140000  # the `sustained` leg: timed like the K steps (barrier + synchronize on both sides); >= 2 s of back-to-back launches
MARG_SUSTAINED_STEPS = 15000   # the marginalised sampler's sustained leg (>= 2 s at ~0.14 ms per step)
PROFILE_TAG = "r05"     # profiles/<tag>_summary.json: rocprofv3 PMC passes of this command (tools/profile_round.sh)
MARG_K = MARG_Q = 4     # marginalised leg: sub-steps per EEP interval x mass ratios


def launch_ranks(args) -> int:
    """Plain `bench.py --gpus N`: start N rank processes (no GPU call has been made in this one), relay rank 0's line.
    Exit code: the FIRST failure seen (peers killed afterwards die of our SIGTERM, which says nothing); 124 when a
    deadline ended the launch."""
    dist_dir = tempfile.mkdtemp(prefix="b9dist_")
    nonce = f"{os.getpid()}_{time.time_ns()}"

    def seconds(name, default):
        try:
            v = float(os.environ.get(name, ""))
            return v if v >= 0.0 else default
        except ValueError:
            return default
    init_deadline, run_deadline = seconds("B9_LAUNCH_TIMEOUT_S", 300.0), seconds("B9_RUN_TIMEOUT_S", 0.0)
    procs, logs = [], []
    first_fail, live, kill_at = 0, {}, None
    got_signal = []
    # SIGINT / SIGTERM to the launcher: forwarded to the ranks (SIGTERM, SIGKILL 5 s later) by the wait loop; the finally
    # block below reaps whatever is left and removes the bootstrap directory on every way out
    old_handlers = {sig: signal.signal(sig, lambda s_, f_: got_signal.append(s_)) for sig in (signal.SIGINT, signal.SIGTERM)}

    def end_all(sig):
        for q in live.values():
            q.send_signal(sig)

    def die_with_parent():       # (child side, before exec) a launcher killed without a word still ends its ranks
        try:
            import ctypes
            ctypes.CDLL(None).prctl(1, int(signal.SIGTERM))     # PR_SET_PDEATHSIG
        except Exception:
            pass
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_RANK=str(r), B9_DIST_DIR=dist_dir,
                       MASTER_ADDR="127.0.0.1", B9_LAUNCH_NONCE=nonce, B9_LAUNCHER_OWNS_DIR="1")
            if args.force_ranks:
                env["B9_FORCE_RANKS"] = "1"
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            out = None if r == 0 else open(os.path.join(dist_dir, f"rank{r}.log"), "w")
            logs.append(out)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out, stderr=subprocess.STDOUT if out else None, preexec_fn=die_with_parent))
        # wait for all of them; a rank that fails takes the others with it (a peer blocked in the communicator would wait
        # for it for ever) -- exactly the processes started above, by pid
        t0 = time.monotonic()
        live, all_ready = dict(enumerate(procs)), False
        while live:
            if got_signal and kill_at is None:
                sys.stderr.write("bench.py launcher: signal %d: ending the %d rank(s)\n" % (got_signal[0], len(live)))
                first_fail = first_fail or 128 + int(got_signal[0])
                end_all(signal.SIGTERM)
                kill_at = time.monotonic()
            for r, p in list(live.items()):
                code = p.poll()
                if code is None:
                    continue
                del live[r]
                if code != 0 and not first_fail:
                    first_fail = code if code > 0 else 128 - code
                if code != 0 and kill_at is None:
                    end_all(signal.SIGTERM)
                    kill_at = time.monotonic()
            if not live:
                break
            if not all_ready:
                all_ready = all(os.path.exists(os.path.join(dist_dir, f"ready.{nonce}.{r}")) for r in range(args.gpus))
            late_start = not all_ready and init_deadline > 0 and time.monotonic() - t0 > init_deadline
            late_run = run_deadline > 0 and time.monotonic() - t0 > run_deadline
            if (late_start or late_run) and kill_at is None:
                sys.stderr.write("bench.py launcher: %s after %.0f s (%s): ending the %d remaining rank(s)\n" % (
                    "not every rank brought its RCCL communicator up" if late_start else "the run did not finish",
                    time.monotonic() - t0, "B9_LAUNCH_TIMEOUT_S" if late_start else "B9_RUN_TIMEOUT_S", len(live)))
                first_fail = first_fail or 124
                end_all(signal.SIGTERM)
                kill_at = time.monotonic()
            if kill_at is not None and time.monotonic() - kill_at > 5.0:
                end_all(signal.SIGKILL)
            time.sleep(0.02)
    finally:
        for p in procs:          # any way out (an OSError while starting ranks, an exception in the loop): no rank is left behind
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=5.0)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for r, f in enumerate(logs):
            if f:
                f.close()
                txt = open(f.name).read()
                if txt.strip() and first_fail:
                    sys.stderr.write(f"---- rank {r} ----\n{txt}\n")
                os.unlink(f.name)
        for name in [f"rccl_id.{nonce}", f"rccl_id.{nonce}.tmp"] + [f"ready.{nonce}.{r}" for r in range(args.gpus)]:
            try:
                os.unlink(os.path.join(dist_dir, name))
            except OSError:
                pass
        try:
            os.rmdir(dist_dir)
        except OSError:
            pass
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
    return first_fail


def check_group(world: int, rccl_ranks: int, devices) -> None:
    """A multi-rank line must come from `world` ranks on `world` DISTINCT GPUs: exits non-zero (3) otherwise.  The verdict
    is the C++ library's (b9h::group_error), the same one the RCCL exchange applies to itself at start-up."""
    import ctypes as C
    from base_amd import hostlib
    msg = C.create_string_buffer(512)
    if hostlib.load().b9h_group_check(int(world), int(rccl_ranks), ",".join(devices).encode(), msg, len(msg)) != 0:
        sys.stderr.write("bench.py: " + msg.value.decode() + "\n")
        raise SystemExit(3)


# ---- algorithmic fp64 work per star-eval ("Algorithmic operations") ---------------------------------
# One lane-operation = one fp64 add / multiply / fma / compare of one lane.  A division, an exponential and a logarithm
# are priced at the instruction count of the leanest 1-ulp implementation this repo has (rcp + Newton 10, Cody-Waite +
# Horner exp 20, fdlibm-polynomial log 35): they are operations the MATH needs, whatever the code does.
OP_DIV, OP_EXP, OP_LOG = 10, 20, 35


def useful_lane_ops(n_filt: int, n_eep: int, frac_binary: float, n_pops: int = 1) -> dict:
    rounds = max(1, math.ceil(math.log(max(n_eep, 2)) / math.log(8.0)))
    search = 7 * rounds                     # 8-ary bracket search: 7 compares per round
    weight = 2 + OP_DIV                     # t = (m - mass[lo]) / (mass[lo+1] - mass[lo])
    lerp = 2 * n_filt                       # per filter: b - a, fma
    component = search + weight + lerp
    tail = n_filt + 3 * n_filt + 2          # + modulus/absorption shift; chi^2: sub, mul, fma; c0 - chi^2 / 2
    mixture = OP_EXP + 2                    # product-form field-star mixture: exp, add, multiply (one log per WAVE: not counted)
    combine = n_filt * (2 + OP_EXP + 1 + OP_LOG + 1)    # per filter: (p2 - p1) * k, exp, 1 +, log, fma
    single = n_pops * (component + tail) + mixture + (n_pops - 1) * (2 * OP_EXP + OP_LOG)
    binary = n_pops * (2 * component + combine + tail) + mixture + (n_pops - 1) * (2 * OP_EXP + OP_LOG)
    return {"single": single, "binary": binary, "mean": (1.0 - frac_binary) * single + frac_binary * binary,
            "frac_binary": frac_binary, "search_rounds": rounds,
            "prices": {"div": OP_DIV, "exp": OP_EXP, "log": OP_LOG, "add_mul_fma_cmp": 1}}


def profile_counters(prefix: str, source_hash: str):
    """Per-launch counters of the kernel whose name starts with `prefix` from the committed rocprofv3 PMC passes of this
    command, with their provenance -- or {"stale": reason} when they were measured on other kernel sources."""
    pth = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_summary.json")
    if not os.path.exists(pth):
        return {"stale": f"profiles/{PROFILE_TAG}_summary.json is missing"}
    doc = json.load(open(pth))
    if doc.get("csrc_sha256") != source_hash:
        return {"stale": f"profiles/{PROFILE_TAG}_summary.json was measured on other kernel sources (csrc_sha256 "
                         f"{str(doc.get('csrc_sha256'))[:12]} != this tree's {source_hash[:12]}): counters dropped",
                "source": f"profiles/{PROFILE_TAG}_summary.json", "commit": doc.get("commit")}
    # (several instances may share the prefix -- the marginalised catalogue plan's one-launch counting pass is a k_star_marg
    #  too: the instance the run spent most time in is the one meant)
    names = [k for k in doc.get("pmc", {}) if k.startswith(prefix)]
    names.sort(key=lambda k: -(doc.get("kernels", {}).get(k, {}).get("calls", 0) * doc.get("kernels", {}).get(k, {}).get("avg_us", 0.0)))
    for kname in names[:1]:
        c = doc["pmc"][kname]
        if True:
            return {"source": f"profiles/{PROFILE_TAG}_summary.json", "commit": doc.get("commit"), "csrc_sha256": doc.get("csrc_sha256"),
                    "command": doc.get("command"), "kernel": kname, "hbm_bytes_per_launch": c.get("hbm_bytes_per_launch"),
                    "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "valu_active_quad_cycles_per_launch": c.get("SQ_ACTIVE_INST_VALU"),
                    "wave_quad_cycles_per_launch": c.get("SQ_WAVE_CYCLES"), "waves_per_launch": c.get("SQ_WAVES"),
                    "lds_bank_conflict_cycles_per_launch": c.get("SQ_LDS_BANK_CONFLICT"),
                    "profiled_avg_us": doc.get("kernels", {}).get(kname, {}).get("avg_us")}
    return {"stale": f"no kernel {prefix}* in profiles/{PROFILE_TAG}_summary.json"}


def valu_roofline(pc, launch_s, useful_ops_per_launch=None):
    """fp64 VALU-issue roofline of one kernel: occupancy fraction from the counters (when they belong to this build),
    useful-work fraction from the algorithmic operation count (always), both over 1024 SIMDs x 2.4 GHz."""
    issue_peak = N_SIMD * CLOCK_GHZ * 1e9                   # VALU issue cycles per second, whole chip
    lane_peak = issue_peak / 4.0 * 64.0                     # fp64 lane-operations per second (16 lanes per SIMD per cycle)
    fresh = bool(pc) and "stale" not in pc
    cyc = 4.0 * pc["valu_active_quad_cycles_per_launch"] if fresh and pc.get("valu_active_quad_cycles_per_launch") else None
    hbm_bytes = pc.get("hbm_bytes_per_launch") if fresh else None
    ok = launch_s and launch_s > 0
    rate = cyc / launch_s if cyc and ok else None
    out = {"bound": "valu", "unit": "fp64 VALU issue cycles/s (all SIMDs)", "achieved": rate, "peak": issue_peak,
           "frac": rate / issue_peak if rate else None, "traffic": hbm_bytes,
           "hbm": {"achieved": hbm_bytes / launch_s / 1e9 if hbm_bytes and ok else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": hbm_bytes / launch_s / 1e9 / HBM_PEAK_GBS if hbm_bytes and ok else None, "bytes_per_launch": hbm_bytes},
           "counters": pc}
    if useful_ops_per_launch is not None and ok:
        out["useful_lane_ops_per_launch"] = useful_ops_per_launch
        out["useful_frac"] = useful_ops_per_launch / launch_s / lane_peak
        out["useful_peak_lane_ops_per_s"] = lane_peak
        if fresh and pc.get("valu_insts_per_launch"):
            out["useful_over_issued_valu"] = useful_ops_per_launch / 64.0 / pc["valu_insts_per_launch"]
    return out


def _delta(got, want, tol=1e-9):
    import numpy as np
    fin = np.isfinite(want)
    rel = np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
    return {"n": int(want.size), "n_finite": int(fin.sum()), "same_support": bool(np.array_equal(np.isfinite(got), fin)),
            "max_abs": float(np.max(np.abs(got[fin] - want[fin]))) if fin.any() else None,
            "max_rel": float(rel.max()) if fin.any() else None, "median_rel": float(np.median(rel)) if fin.any() else None,
            "tolerance_rel": tol, "within_tolerance": bool(fin.any() and rel.max() <= tol and np.array_equal(np.isfinite(got), fin)),
            "against": "this repo's CPU oracle on all host cores (BASE-9 parity unpinned)"}


def native_oracle(pack, stars, priors, options):
    import oracle
    try:
        oracle.build(native=True)
        native = True
    except Exception:
        native = False
    orc = oracle.Oracle(pack, stars, priors, options, native=native)
    # a one-GPU job's CPU share on the GPU boxes is 16 cores, whatever the host exposes
    cores = max(1, min(int(orc.lib.b9o_max_threads()), len(os.sched_getaffinity(0)), 16))
    return orc, cores


def cpu_baseline(pack_d, cl, truth, budget_s: float = 14.0, eng=None, sampler_state=None):
    """Time the CPU oracle on a bounded sample of the same workload (rank 0 only): all host cores
    (OpenMP over stars, as the reference's thread pool [RECALL]) and one thread.  With `eng`, the same
    leg also reports BASELINE.json's second figure, |delta logPost| of the HIP path against that CPU
    path: b9_logpost over 128 random in-grid parameter rows on the full 50k-star cluster, and -- the
    timed path itself -- the sampler's ensemble state (k_mcmc_step's own log-posteriors) after the timed region."""
    import numpy as np
    from base_amd import abi, synth
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors, options = synth.default_priors(pack_d, truth), abi.make_options()
    orc, cores = native_oracle(pack, stars, priors, options)
    params = synth.walker_params(truth, WALKERS_PER_GPU)

    def timed(threads, budget):
        orc.lib.b9o_set_threads(threads)
        orc.logpost(params[:1])                       # warm
        t0 = time.perf_counter()
        reps = 0
        while True:
            orc.logpost(params)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget or reps >= 400:
                break
        return reps * N_STARS * WALKERS_PER_GPU / dt, reps, dt

    v_all, reps, dt = timed(cores, budget_s / 2)
    v_one, reps1, dt1 = timed(1, budget_s / 2)
    best, best_cores = (v_all, cores) if v_all >= v_one else (v_one, 1)      # the CPU's best effort is the baseline
    delta = delta_sampler = None
    orc.lib.b9o_set_threads(cores)
    if eng is not None:
        rows = synth.walker_params(truth, 128, seed=4242, scale=1.0)      # a wide ball around the truth, all inside the grid
        want = orc.logpost(rows)
        got = np.concatenate([eng.logpost(rows[k:k + 32]) for k in range(0, 128, 32)])
        delta = _delta(got, want)
        delta["what"] = "b9_logpost (k_derive_iso + k_star_like + k_finalize) on 128 random in-grid rows, full cluster"
    if sampler_state is not None:
        # the TIMED path's own numbers: the log-posterior every local walker holds after the timed steps was formed by
        # k_mcmc_step (per-wave partials, first-wave decision); the oracle evaluates the same positions
        got = np.asarray(sampler_state["all_logpost"], dtype=np.float64)
        want = orc.logpost(np.asarray(sampler_state["all_params"], dtype=np.float64))
        delta_sampler = _delta(got, want)
        delta_sampler["what"] = ("the sampler's ensemble state after the timed region: log-posteriors formed by k_mcmc_step "
                                 "(the timed kernel) at the walkers' positions, full cluster")
    return {"delta_logpost": delta, "delta_logpost_sampler": delta_sampler,
            "value": best, "unit": "star-likelihood evals/s", "cores": best_cores, "kind": "port",
            "value_all_cores": v_all, "host_cores": cores, "value_1thread": v_one,
            "sample": f"{reps} x logpost of {WALKERS_PER_GPU} walkers x {N_STARS} stars x {N_FILT} filters on {cores} "
                      f"OpenMP thread(s) ({dt:.1f} s) and {reps1} x on 1 thread ({dt1:.1f} s); oracle/b9_oracle.c "
                      f"-O3 -march=native -fopenmp; BASE-9 itself is not mounted: build's CPU oracle, parity unpinned"}


def marg_stats(source_hash: str):
    """Executed / live term counts of k_star_marg per star-eval from the committed stats pass (tools/marg_stats.py ->
    profiles/<tag>_marg_stats.json: a -DB9_MARG_STATS build counting on this workload) -- or {"stale": reason}."""
    pth = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_marg_stats.json")
    if not os.path.exists(pth):
        return {"stale": f"profiles/{PROFILE_TAG}_marg_stats.json is missing"}
    doc = json.load(open(pth))
    if doc.get("csrc_sha256") != source_hash:
        return {"stale": f"profiles/{PROFILE_TAG}_marg_stats.json was counted on other kernel sources: dropped", "source": pth}
    return {"source": f"profiles/{PROFILE_TAG}_marg_stats.json", "live_terms_per_star_eval": doc["live_terms_per_star_eval"],
            "terms_evaluated_per_star_eval": doc["terms_evaluated_per_star_eval"], "live_share": doc["live_share"]}


def marg_useful_lane_ops(n_filt: int, live_terms: float) -> dict:
    """Algorithmic fp64 lane-operations of one marginalised star-eval: every term that ENTERS the star's sum (within 40 e-folds
    of its largest: the committed stats pass counts them) needs its chi^2 (per filter: subtract, multiply, fma), the compare
    that admits it, one exponential and one add; the star then needs one logarithm, and one exp + log for the field-star
    mixture.  Terms a wave evaluates for lanes that do not need them, box tests and table building are NOT useful work."""
    per_term = 3 * n_filt + 1 + OP_EXP + 1
    per_star = OP_LOG + OP_EXP + OP_LOG + 4
    return {"per_live_term": per_term, "per_star": per_star, "live_terms_per_star_eval": live_terms,
            "mean": live_terms * per_term + per_star}


def marginalised_leg(pack, stars, priors, truth, local_rank, source_hash, with_cpu: bool, n_calls: int = 20, steps: int = 400, sustained: bool = True):
    """The marginalised evaluation mode as a measured configuration of its own (every star integrated over primary mass and
    mass ratio, 4 sub-steps per EEP interval x 4 mass ratios = 6384 nodes/star; one LANE per star, a wave walks the node
    table for 64 photometric neighbours):  (i) b9_logpost calls through the C ABI;  (ii) the SAMPLER in this mode -- the C++
    b9h::WalkerSampler driving device-resident blocks, ONE launch per step (k_marg_step: decision + stars + both candidate
    node tables of the next step) -- as MCMC steps/s over `steps` steps and over a sustained leg of >= 2 s with the observed
    shader clock, the ensemble state it ends on checked against the CPU oracle;  (iii) the fp64 VALU rooflines of k_star_marg
    (the b9_logpost calls) and k_marg_step (the sampler), launch times measured live with HIP events: issue occupancy from
    the committed counters and the USEFUL fraction from the committed count of terms that enter a star's sum;  (iv) the CPU
    oracle's brute-force marginalisation on a bounded sample."""
    import numpy as np
    from base_amd import abi, engine, hostlib, mcmc, synth
    opts = abi.make_options(abi.MODE_MARGINALISED, 1, MARG_K, MARG_Q)
    eng = engine.Engine(pack, stars, priors, opts, device=local_rank)
    params = synth.walker_params(truth, WALKERS_PER_GPU, seed=43, scale=0.02)
    eng.logpost(params)
    eng.enable_timing(1)
    eng.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(n_calls):
        eng.logpost(params)
    dt = (time.perf_counter() - t0) / n_calls
    k_ms, k_n = eng.kernel_time_ms(reset=True)
    eng.enable_timing(0)
    launch_s = k_ms / max(k_n, 1) * 1e-3
    nodes = (eng.max_eep() - 1) * MARG_K * MARG_Q
    evals = N_STARS * WALKERS_PER_GPU
    out = {"value": evals / dt, "unit": "star-likelihood evals/s", "iso_increm": MARG_K, "n_q": MARG_Q,
           "nodes_per_star_eval": nodes, "node_evals_per_s": evals * nodes / dt,
           "ms_per_logpost_call": 1e3 * dt, "calls": n_calls,
           "config": {"workload": f"the bench cluster ({N_STARS} stars x {N_FILT} filters x {WALKERS_PER_GPU} walkers), marginalised mode: "
                                  f"{MARG_K} sub-steps per EEP interval x {MARG_Q} mass ratios; b9_logpost calls (derive + k_marg_table + k_star_marg + finalize)"}}
    # ---- the sampler in this mode: the same C++ driver as the headline, one launch per step
    free = mcmc.DEFAULT_FREE
    smp = hostlib.HostSampler(WALKERS_PER_GPU, free, [mcmc.DEFAULT_STEP[k] for k in free], hostlib.Exchange.local(), seed=2025,
                              block=MCMC_BLOCK, engine=eng)
    smp.initialise(synth.walker_params(truth, WALKERS_PER_GPU, seed=42, scale=0.02))
    smp.run(200)                                            # untimed: two adaptation blocks
    acc0 = smp.state()["accepted_local"]
    hostlib._check(hostlib.load().b9h_device_synchronize())
    t0 = time.perf_counter()
    smp.run(steps)
    hostlib._check(hostlib.load().b9h_device_synchronize())
    dt_s = time.perf_counter() - t0
    st = smp.state()
    out["sampler"] = {"mcmc_steps_per_s": steps / dt_s, "ms_per_step": 1e3 * dt_s / steps, "steps": steps, "warmup_steps_untimed": 200,
                      "value": evals * steps / dt_s, "unit": "star-likelihood evals/s", "walkers": WALKERS_PER_GPU, "mcmc_block": MCMC_BLOCK,
                      "accept_rate": (st["accepted_local"] - acc0) / float(WALKERS_PER_GPU * steps),
                      "driver": "C++ host library (b9h::WalkerSampler), device-resident blocks; a step = ONE k_marg_step launch (the previous "
                                "step's decision, the stars against the chosen candidate's node table, both candidate tables of the next step)"}
    stats = marg_stats(source_hash)
    useful = marg_useful_lane_ops(N_FILT, stats["live_terms_per_star_eval"]) if "stale" not in stats else None
    # ... and its sustained leg: >= 2 s of back-to-back launches, the shader clock stamped at both ends, every 50th launch
    # of k_marg_step opening a HIP-event bracket over 8 launches
    if sustained:
        eng.enable_timing(2 * TIMING_EVERY)                 # (300 brackets: inside the pre-created event pool)
        eng.kernel_time_ms(reset=True)
        hostlib._check(hostlib.load().b9h_device_synchronize())
        t0 = time.perf_counter()
        eng.clock_stamp(0)
        smp.run(MARG_SUSTAINED_STEPS)
        eng.clock_stamp(1)
        hostlib._check(hostlib.load().b9h_device_synchronize())
        dt_ss = time.perf_counter() - t0
        ks_ms, ks_n = eng.kernel_time_ms(reset=True)
        eng.enable_timing(0)
        clock = eng.clock_mhz()
        step_s = ks_ms / max(ks_n, 1) * 1e-3
        roof_s = valu_roofline(profile_counters("k_marg_step", source_hash), step_s if ks_n else None, useful["mean"] * evals if useful else None)
        roof_s.update({"kernel": "k_marg_step", "launches_timed": ks_n, "avg_launch_us": 1e6 * step_s if ks_n else None, "star_evals_per_launch": evals,
                       "note": "the sampler's launch: decision + 6256 star workgroups + 408 front workgroups (writers, table builders); launch period "
                               "from HIP-event brackets over 8 launches in the sustained leg; useful work = the star role's (the builders' tables "
                               "are overhead of the speculation, not counted)"})
        if clock["mhz"] > 0:
            scale = CLOCK_GHZ * 1e3 / clock["mhz"]
            roof_s["frac_at_observed_clock"] = roof_s["frac"] * scale if roof_s.get("frac") is not None else None
            roof_s["useful_frac_at_observed_clock"] = roof_s["useful_frac"] * scale if roof_s.get("useful_frac") is not None else None
        out["sampler"]["sustained"] = {"steps": MARG_SUSTAINED_STEPS, "seconds": dt_ss, "ms_per_step": 1e3 * dt_ss / MARG_SUSTAINED_STEPS,
                                       "mcmc_steps_per_s": MARG_SUSTAINED_STEPS / dt_ss, "value": evals * MARG_SUSTAINED_STEPS / dt_ss,
                                       "unit": "star-likelihood evals/s", "clock": clock}
        out["sampler"]["clock_mhz_observed"] = clock["mhz"]
        out["sampler"]["roofline"] = roof_s
    st = smp.state()
    roof = valu_roofline(profile_counters("k_star_marg", source_hash), launch_s if k_n else None,
                         useful["mean"] * evals if useful else None)
    roof.update({"kernel": "k_star_marg", "launches_timed": k_n, "avg_launch_us": 1e6 * launch_s if k_n else None,
                 "star_evals_per_launch": evals, "useful_ops_per_star_eval": useful, "term_counts": stats,
                 "algorithmic_bytes_per_launch": evals * 152.0 / WALKERS_PER_GPU + WALKERS_PER_GPU * eng.max_eep() * (N_FILT + 1) * 8.0,
                 "note": "launch time measured live (HIP events around the k_marg_table + k_star_marg pair of every call); frac = issue-slot "
                         "occupancy from the committed PMC pass; useful_frac = (terms that enter a star's sum, from the committed stats "
                         "pass, x the lane-operations one such term needs + the star's closing logarithms) x star-evals / launch time / "
                         "(1024 SIMDs x 16 lanes x 2.4 GHz); both null when the kernel sources differ from the measured ones"})
    out["roofline"] = roof
    if with_cpu:
        orc, cores = native_oracle(pack, stars, priors, opts)
        orc.lib.b9o_set_threads(cores)
        row = params[:1]
        t0 = time.perf_counter()
        want_lp, want_ps = orc.logpost(row, perstar=True)                 # ONE row, every star: the bounded sample
        dt_cpu = time.perf_counter() - t0
        got_lp, got_ps = eng.logpost(row, perstar=True)
        d = _delta(np.concatenate([got_lp, got_ps.ravel()]), np.concatenate([want_lp, want_ps.ravel()]))
        d["what"] = "log-posterior of one row + all 50 000 per-star values, marginalised mode"
        out["cpu_baseline"] = {"value": N_STARS / dt_cpu, "unit": "star-likelihood evals/s", "cores": cores, "kind": "port",
                               "sample": f"1 row x {N_STARS} stars x {nodes} nodes on {cores} OpenMP thread(s) ({dt_cpu:.1f} s); "
                                         f"oracle/b9_oracle.c star_marg_loglike (brute force over the whole grid, no pruning)",
                               "delta_logpost": d}
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        # the sampler's own numbers: the log-posteriors its walkers hold after the timed steps (formed by k_marg_step's launches);
        # two of the eight walkers (the brute-force oracle needs ~4 s per row)
        pick = [0, WALKERS_PER_GPU - 1]
        got = np.asarray(st["all_logpost"], dtype=np.float64)[pick]
        want = orc.logpost(np.asarray(st["all_params"], dtype=np.float64)[pick])
        ds = _delta(got, want)
        ds["what"] = ("the marginalised sampler's ensemble state after its sustained leg: log-posteriors formed by k_marg_step's launches "
                      f"at the positions of walkers {pick[0]} and {pick[1]}, full cluster, against the oracle's brute-force integral")
        out["sampler"]["delta_logpost_sampler"] = ds
    smp.close()
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--force-ranks", action="store_true",
                    help="with --gpus 1: take the multi-rank route anyway (self-launched child, RCCL communicator of one rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-marginalised", action="store_true", help="diagnostic: skip the marginalised-mode leg")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: skip the HIP-event bracketing of k_mcmc_step")
    ap.add_argument("--no-prewarm", action="store_true", help="diagnostic: skip the untimed pre-warm block")
    ap.add_argument("--no-sustained", action="store_true", help="diagnostic: skip the sustained legs")
    args = ap.parse_args()

    is_rank = "RANK" in os.environ or "B9_RANK" in os.environ
    if (args.gpus > 1 or args.force_ranks) and not is_rank:
        raise SystemExit(launch_ranks(args))          # (nothing above touches a GPU)

    import numpy as np
    from base_amd import abi, build, engine, hostlib, mcmc, synth

    rank, world, local_rank = hostlib.rank_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    hostlib.load().b9h_test_stall(b"start", rank)       # (test hook of the launcher's deadline; no GPU call yet)

    # ---- synthetic inputs (identical on every rank: fixed seeds) -----------------------------
    pack_d = synth.make_pack("parsec", N_FILT)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, N_STARS, seed=9003, truth=truth)
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors, options = synth.default_priors(pack_d, truth), abi.make_options()
    eng = engine.Engine(pack, stars, priors, options, device=local_rank)
    use_rccl = world > 1 or bool(hostlib.load().b9h_forced_ranks())
    exchange = hostlib.Exchange.rccl(rank, world, eng.device_id()) if use_rccl else hostlib.Exchange.local()
    n_walkers = WALKERS_PER_GPU * world
    start = synth.walker_params(truth, n_walkers, seed=42, scale=0.02)
    free = mcmc.DEFAULT_FREE
    sampler = hostlib.HostSampler(n_walkers, free, [mcmc.DEFAULT_STEP[k] for k in free], exchange, seed=2024,
                                  block=MCMC_BLOCK, engine=eng)
    sampler.initialise(start)

    def barrier():
        exchange.barrier()
        hostlib._check(hostlib.load().b9h_device_synchronize())

    prewarm = 0 if args.no_prewarm else PREWARM_STEPS
    if prewarm:
        sampler.run(prewarm)                # untimed and not part of W: clocks, first touch, RCCL channels
    sampler.run(args.warmup)                # the W untimed warm-up steps
    acc0 = sampler.state()["accepted_local"]
    eng.enable_timing(0 if args.no_kernel_timing else TIMING_EVERY)     # also pre-creates the HIP-event pool
    eng.kernel_time_ms(reset=True)
    barrier()
    t0 = time.perf_counter()
    sampler.run(args.steps)                 # exactly K steps, in device-resident blocks of <= MCMC_BLOCK (one C++ call)
    barrier()
    dt_local = time.perf_counter() - t0
    k_ms, k_n = eng.kernel_time_ms(reset=True)
    eng.enable_timing(0)
    bracket_ms = eng.calibrate_timing()     # event bracket around an empty kernel, same stream
    dt = exchange.max(dt_local)             # max over ranks (RCCL all-reduce)
    st = sampler.state()                    # (the state the parity check below reads: right after the K timed steps)
    if world > 1:
        check_group(world, exchange.comm_ranks, exchange.devices)
    # ---- sustained leg: the same loop over 140 000 steps (>= 2 s), timed the same way, with the shader clock stamped on the stream at
    # both ends (in-kernel s_memtime against the 100 MHz s_memrealtime: what the chip ran at, not what sysfs says)
    sustained = None
    if not args.no_sustained:
        barrier()
        t0 = time.perf_counter()
        eng.clock_stamp(0)
        sampler.run(SUSTAINED_STEPS)
        eng.clock_stamp(1)
        barrier()
        dts = exchange.max(time.perf_counter() - t0)
        sustained = {"steps": SUSTAINED_STEPS, "ms_per_step": 1e3 * dts / SUSTAINED_STEPS, "mcmc_steps_per_s": SUSTAINED_STEPS / dts,
                     "value": float(N_STARS) * WALKERS_PER_GPU * world * SUSTAINED_STEPS / dts, "unit": "star-likelihood evals/s",
                     "seconds": dts, "clock": eng.clock_mhz(),
                     "note": "same sampler, same blocks, directly after the K timed steps; timed between barrier + device synchronize "
                             "on both sides, max over ranks; the two clock stamps are inside the timed stretch"}

    if rank == 0:
        source_hash = build.source_hash()
        evals = float(N_STARS) * n_walkers * args.steps
        star_evals_launch = float(N_STARS) * WALKERS_PER_GPU
        # The HIP-event bracket spans 8 consecutive launches; its time / 8 is the kernel's launch PERIOD (duration +
        # the ~1.5 us dispatch boundary), a little above rocprofv3's kernel-only average of the same command (profiles/).
        k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
        ops = useful_lane_ops(N_FILT, eng.max_eep(), float(np.mean(np.asarray(cl["mass_ratio"]) > 0.0)))
        roof = valu_roofline(profile_counters("k_mcmc_step", source_hash), k_avg_s if k_n else None, ops["mean"] * star_evals_launch)
        alg152, alg_layout = 152.0, float(eng.bytes_per_star_eval())
        kernel_ms = 1e3 * k_avg_s * args.steps if k_n else None
        roof.update({"kernel": "k_mcmc_step", "launches_timed": k_n, "timed_every": TIMING_EVERY, "launches_per_bracket": 8,
                     "avg_launch_us": 1e6 * k_avg_s, "empty_kernel_bracket_us": 1e3 * bracket_ms,
                     "useful_ops_per_star_eval": ops,
                     "algorithmic": {"bytes_per_star_eval_survey_8d": alg152, "bytes_per_star_eval_layout": alg_layout,
                                     "star_evals_per_launch": star_evals_launch,
                                     "rate_GBps_8d": star_evals_launch * alg152 / k_avg_s / 1e9 if k_n else None,
                                     "frac_of_hbm_peak_8d": star_evals_launch * alg152 / k_avg_s / 1e9 / HBM_PEAK_GBS if k_n else None,
                                     "note": "an L2-served rate: the 8 walkers of a GPU re-read a star tile from the XCD-local L2, "
                                             "so these bytes never cross HBM 8 times; NOT an HBM fraction"},
                     "note": "launch duration measured live (HIP events on the launch stream); frac = issue-slot occupancy from the "
                             "committed rocprofv3 PMC passes of this command (counters.source / .commit; null when the kernel sources "
                             "differ from the profiled ones); useful_frac = algorithmic fp64 lane-operations (DESIGN.md section 3) "
                             "x star-evals / launch time / (1024 SIMDs x 16 lanes x 2.4 GHz) -- needs no counter"})
        if sustained and sustained["clock"]["mhz"] > 0:
            scale = CLOCK_GHZ * 1e3 / sustained["clock"]["mhz"]
            roof["frac_at_observed_clock"] = roof["frac"] * scale if roof.get("frac") is not None else None
            roof["useful_frac_at_observed_clock"] = roof["useful_frac"] * scale if roof.get("useful_frac") is not None else None
        out = {
            "metric": "star-likelihood evals/sec", "value": evals / dt, "unit": "star-likelihood evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "mcmc_steps_per_s": args.steps / dt,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] (C2 in DESIGN.md): 50k-star x 8-filter synthetic cluster, PARSEC-shaped synthetic pack "
                                   "(10 FeH x 60 ages x 400 EEPs), given-mass mode, 8 walkers per GPU; one LANE per star "
                                   "(64-star chunks, per-wave shuffle reduction, fixed-order sum of the per-wave partials) -- "
                                   "not one wavefront per star: with one interpolation per star a wave per star would idle 63 "
                                   "lanes; the marginalised mode (marginalised_mode below) is one lane per star too (a wave walks the node "
                                   "table for its 64 stars)",
                       "n_stars": N_STARS, "n_filters": N_FILT, "walkers_per_gpu": WALKERS_PER_GPU,
                       "walkers_total": n_walkers, "parallelism": f"walkers{world}", "ranks": exchange.world,
                       "rccl_ranks": exchange.comm_ranks, "devices": exchange.devices, "forced_ranks": bool(args.force_ranks),
                       "mcmc_block": MCMC_BLOCK, "driver": "C++ host library (b9h::WalkerSampler), one call for the K steps",
                       "collective": (exchange.name + "; one all-gather of [logpost, position, moments] rows per block") if use_rccl else "none",
                       "wd_tracks": "rectangular cooling table (wc_uniform path); the bench cluster has no WD-stage stars -- the ragged-track "
                                    "path of real cooling models is timed in profiles/ (config sweep, row C3r)",
                       "prewarm_steps_untimed": prewarm,
                       "steps_per_launch": eng.step_depth(WALKERS_PER_GPU),
                       "steps_per_launch_note": "Metropolis steps of every chain per k_mcmc_step / k_mcmc_tree launch: 1 at 8 walkers x 50k stars (the "
                                                "launch is full); the single-chain BASELINE shapes run the tree-speculative launch, 3 steps each "
                                                "(profiles/ config sweep)"},
            "roofline": roof,
            "timed_region_breakdown": {"kernel_ms": kernel_ms, "host_and_block_fixed_ms": (1e3 * dt - kernel_ms) if kernel_ms else None,
                                       "wall_ms": 1e3 * dt, "ms_per_step_over_launch_period": (1e3 * dt / args.steps) / (1e3 * k_avg_s) if k_n else None},
            "accept_rate": (st["accepted_local"] - acc0) / float(WALKERS_PER_GPU * args.steps),
            "sustained": sustained,
            "clock_mhz_observed": sustained["clock"]["mhz"] if sustained else None,
            "clock_note": "shader clock over the sustained leg (median over compute units of delta s_memtime / delta s_memrealtime x 100 MHz); the "
                          "roofline peaks are quoted at the 2.4 GHz maximum -- roofline.frac_at_observed_clock rescales them",
            "parity": "vs this repo's CPU oracle (BASE-9 parity unpinned: reference source not mounted)",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pack_d, cl, truth, eng=eng, sampler_state=st)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if world == 1 and not args.no_marginalised:
            out["marginalised_mode"] = marginalised_leg(pack, stars, priors, truth, local_rank, source_hash,
                                                        with_cpu=not args.no_cpu_baseline, sustained=not args.no_sustained)
        print(json.dumps(out), flush=True)
    exchange.barrier()
    sampler.close()
    exchange.close()
    eng.close()


if __name__ == "__main__":
    main()
