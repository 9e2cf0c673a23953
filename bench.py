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
anything touches a GPU, waits for them and relays rank 0's JSON line.  Under a launcher that already exports RANK /
WORLD_SIZE / LOCAL_RANK (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...) the process is a rank.

value        = n_stars x total walkers x K / max-over-ranks wall time   (whole job, all GPUs)
roofline     = the dominant kernel (k_mcmc_step, the fused sampler step).  The kernel is NOT HBM-bound: the walkers of
               a GPU share every star tile through the XCD-local L2, and its waves spend their time in dependent
               fp64 VALU chains.  Three figures are reported, each against its own peak: fp64 VALU issue (the bound
               the counters name; `frac`), HBM traffic (counters), and the algorithmic byte rate of SURVEY.md 8(d).
               Launch duration is measured live with HIP events on the launch stream; instruction and byte counts per
               launch come from the committed rocprofv3 PMC passes of this same command (profiles/, tagged).
cpu_baseline = the CPU oracle ("port"; the reference itself is not mounted) timed on this
               box's host cores on a bounded sample of the same workload (rank 0, N=1)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
PROFILE_TAG = "r02"     # profiles/<tag>_summary.json: rocprofv3 PMC passes of this command (tools/profile_round.sh)


def launch_ranks(args) -> int:
    """Plain `bench.py --gpus N`: start N rank processes (no GPU call has been made in this one), relay rank 0's line."""
    dist_dir = tempfile.mkdtemp(prefix="b9dist_")
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_RANK=str(r), B9_DIST_DIR=dist_dir,
                   MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else open(os.path.join(dist_dir, f"rank{r}.log"), "w")
        logs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out, stderr=subprocess.STDOUT if out else None))
    # wait for all of them; a rank that fails takes the others with it (a peer blocked in the communicator's
    # initialisation would wait for it for ever) -- exactly the processes started above, by pid
    rc, live = 0, dict(enumerate(procs))
    while live:
        for r, p in list(live.items()):
            code = p.poll()
            if code is None:
                continue
            del live[r]
            rc = max(rc, abs(code))
            if code != 0:
                for q in live.values():
                    q.terminate()
        if live:
            time.sleep(0.05)
    for r, f in enumerate(logs):
        if f:
            f.close()
            txt = open(f.name).read()
            if txt.strip() and rc:
                sys.stderr.write(f"---- rank {r} ----\n{txt}\n")
            os.unlink(f.name)
    try:
        os.rmdir(dist_dir)
    except OSError:
        pass
    return rc


def cpu_baseline(pack_d, cl, truth, budget_s: float = 14.0, eng=None):
    """Time the CPU oracle on a bounded sample of the same workload (rank 0 only): all host cores
    (OpenMP over stars, as the reference's thread pool [RECALL]) and one thread.  With `eng`, the same
    leg also reports BASELINE.json's second figure, |delta logPost| of the HIP path against that CPU
    path, over 128 random in-grid parameter rows on the full 50k-star cluster."""
    import numpy as np
    import oracle
    from base_amd import abi, synth
    try:
        oracle.build(native=True)
        native = True
    except Exception:
        native = False
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors, options = synth.default_priors(pack_d, truth), abi.make_options()
    orc = oracle.Oracle(pack, stars, priors, options, native=native)
    params = synth.walker_params(truth, WALKERS_PER_GPU)
    # a one-GPU job's CPU share on the GPU boxes is 16 cores, whatever the host exposes
    cores = max(1, min(int(orc.lib.b9o_max_threads()), len(os.sched_getaffinity(0)), 16))

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
    delta = None
    if eng is not None:
        orc.lib.b9o_set_threads(cores)
        rows = synth.walker_params(truth, 128, seed=4242, scale=1.0)      # a wide ball around the truth, all inside the grid
        want = orc.logpost(rows)
        got = np.concatenate([eng.logpost(rows[k:k + 32]) for k in range(0, 128, 32)])
        fin = np.isfinite(want)
        rel = np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))
        delta = {"n_rows": 128, "n_finite": int(fin.sum()), "same_support": bool(np.array_equal(np.isfinite(got), fin)),
                 "max_abs": float(np.max(np.abs(got[fin] - want[fin]))), "max_rel": float(rel.max()), "median_rel": float(np.median(rel)),
                 "tolerance_rel": 1e-9, "against": "this repo's CPU oracle on all host cores (BASE-9 parity unpinned)"}
    return {"delta_logpost": delta, "value": best, "unit": "star-likelihood evals/s", "cores": best_cores, "kind": "port",
            "value_all_cores": v_all, "host_cores": cores, "value_1thread": v_one,
            "sample": f"{reps} x logpost of {WALKERS_PER_GPU} walkers x {N_STARS} stars x {N_FILT} filters on {cores} "
                      f"OpenMP thread(s) ({dt:.1f} s) and {reps1} x on 1 thread ({dt1:.1f} s); oracle/b9_oracle.c "
                      f"-O3 -march=native -fopenmp; BASE-9 itself is not mounted: build's CPU oracle, parity unpinned"}


def marginalised_leg(pack, stars, priors, truth, local_rank, n_calls: int = 5):
    """Secondary figure: the marginalised mode (one wavefront per star; every star integrated over
    primary mass and mass ratio, 4 sub-steps per EEP interval x 4 mass ratios = 6384 nodes/star)."""
    from base_amd import abi, engine, synth
    K = Q = 4
    eng = engine.Engine(pack, stars, priors, abi.make_options(abi.MODE_MARGINALISED, 1, K, Q), device=local_rank)
    params = synth.walker_params(truth, WALKERS_PER_GPU, seed=43, scale=0.02)
    eng.logpost(params)
    t0 = time.perf_counter()
    for _ in range(n_calls):
        eng.logpost(params)
    dt = (time.perf_counter() - t0) / n_calls
    nodes = (eng.max_eep() - 1) * K * Q
    eng.close()
    return {"value": N_STARS * WALKERS_PER_GPU / dt, "unit": "star-likelihood evals/s", "iso_increm": K, "n_q": Q,
            "nodes_per_star_eval": nodes, "node_evals_per_s": N_STARS * WALKERS_PER_GPU * nodes / dt,
            "ms_per_logpost_call": 1e3 * dt, "calls": n_calls}


def profile_counters():
    """Per-launch counters of k_mcmc_step from the committed rocprofv3 PMC passes of this command, with their provenance."""
    pth = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_summary.json")
    if not os.path.exists(pth):
        return None
    doc = json.load(open(pth))
    for kname, c in doc.get("pmc", {}).items():
        if kname.startswith("k_mcmc_step"):
            return {"source": f"profiles/{PROFILE_TAG}_summary.json", "commit": doc.get("commit"), "command": doc.get("command"),
                    "kernel": kname, "hbm_bytes_per_launch": c.get("hbm_bytes_per_launch"),
                    "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "valu_active_quad_cycles_per_launch": c.get("SQ_ACTIVE_INST_VALU"),
                    "wave_quad_cycles_per_launch": c.get("SQ_WAVE_CYCLES"), "waves_per_launch": c.get("SQ_WAVES")}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: skip the HIP-event bracketing of k_mcmc_step")
    ap.add_argument("--no-prewarm", action="store_true", help="diagnostic: skip the untimed pre-warm block")
    args = ap.parse_args()

    is_rank = "RANK" in os.environ or "B9_RANK" in os.environ
    if args.gpus > 1 and not is_rank:
        raise SystemExit(launch_ranks(args))          # (nothing above touches a GPU)

    import numpy as np
    from base_amd import abi, engine, hostlib, mcmc, synth

    rank, world, local_rank = hostlib.rank_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")

    # ---- synthetic inputs (identical on every rank: fixed seeds) -----------------------------
    pack_d = synth.make_pack("parsec", N_FILT)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, N_STARS, seed=9003, truth=truth)
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors, options = synth.default_priors(pack_d, truth), abi.make_options()
    eng = engine.Engine(pack, stars, priors, options, device=local_rank)
    exchange = hostlib.Exchange.rccl(rank, world, eng.device_id()) if world > 1 else hostlib.Exchange.local()
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
    st = sampler.state()

    if rank == 0:
        evals = float(N_STARS) * n_walkers * args.steps
        star_evals_launch = float(N_STARS) * WALKERS_PER_GPU
        # The HIP-event bracket spans 8 consecutive launches; its time / 8 is the kernel's launch PERIOD (duration +
        # the ~1.5 us dispatch boundary), a little above rocprofv3's kernel-only average of the same command (profiles/).
        k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
        pc = profile_counters()
        valu_cycles = 4.0 * pc["valu_active_quad_cycles_per_launch"] if pc and pc.get("valu_active_quad_cycles_per_launch") else None
        valu_peak = N_SIMD * CLOCK_GHZ * 1e9                   # VALU issue cycles per second, whole chip
        valu_rate = valu_cycles / k_avg_s if valu_cycles and k_n else None
        hbm_bytes = pc["hbm_bytes_per_launch"] if pc else None
        hbm_rate = hbm_bytes / k_avg_s / 1e9 if hbm_bytes and k_n else None
        alg152, alg_layout = 152.0, float(eng.bytes_per_star_eval())
        kernel_ms = 1e3 * k_avg_s * args.steps if k_n else None
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
                                   "lanes; the marginalised mode (marginalised_mode below) is one wavefront per star",
                       "n_stars": N_STARS, "n_filters": N_FILT, "walkers_per_gpu": WALKERS_PER_GPU,
                       "walkers_total": n_walkers, "parallelism": f"walkers{world}", "ranks": exchange.world,
                       "mcmc_block": MCMC_BLOCK, "driver": "C++ host library (b9h::WalkerSampler), one call for the K steps",
                       "collective": (exchange.name + "; one all-gather of [logpost, position, moments] rows per block") if world > 1 else "none",
                       "prewarm_steps_untimed": prewarm},
            "roofline": {"bound": "valu", "unit": "fp64 VALU issue cycles/s (all SIMDs)",
                         "achieved": valu_rate, "peak": valu_peak, "frac": (valu_rate / valu_peak) if valu_rate else None,
                         "traffic": hbm_bytes,
                         "kernel": "k_mcmc_step", "launches_timed": k_n, "timed_every": TIMING_EVERY, "launches_per_bracket": 8,
                         "avg_launch_us": 1e6 * k_avg_s, "empty_kernel_bracket_us": 1e3 * bracket_ms,
                         "hbm": {"achieved": hbm_rate, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (hbm_rate / HBM_PEAK_GBS) if hbm_rate else None, "bytes_per_launch": hbm_bytes},
                         "algorithmic": {"bytes_per_star_eval_survey_8d": alg152, "bytes_per_star_eval_layout": alg_layout,
                                         "star_evals_per_launch": star_evals_launch,
                                         "rate_GBps_8d": star_evals_launch * alg152 / k_avg_s / 1e9 if k_n else None,
                                         "frac_of_hbm_peak_8d": star_evals_launch * alg152 / k_avg_s / 1e9 / HBM_PEAK_GBS if k_n else None,
                                         "note": "an L2-served rate: the 8 walkers of a GPU re-read a star tile from the XCD-local L2, "
                                                 "so these bytes never cross HBM 8 times; NOT an HBM fraction"},
                         "counters": pc,
                         "note": "launch duration measured live (HIP events on the launch stream); per-launch instruction and "
                                 "byte counts from the committed rocprofv3 PMC passes of this command (counters.source / .commit)"},
            "timed_region_breakdown": {"kernel_ms": kernel_ms, "host_and_block_fixed_ms": (1e3 * dt - kernel_ms) if kernel_ms else None,
                                       "wall_ms": 1e3 * dt, "ms_per_step_over_launch_period": (1e3 * dt / args.steps) / (1e3 * k_avg_s) if k_n else None},
            "accept_rate": (st["accepted_local"] - acc0) / float(WALKERS_PER_GPU * args.steps),
            "parity": "vs this repo's CPU oracle (BASE-9 parity unpinned: reference source not mounted)",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pack_d, cl, truth, eng=eng)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if world == 1:
            out["marginalised_mode"] = marginalised_leg(pack, stars, priors, truth, local_rank)
        print(json.dumps(out), flush=True)
    exchange.barrier()
    sampler.close()
    exchange.close()
    eng.close()


if __name__ == "__main__":
    main()
