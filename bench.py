#!/usr/bin/env python3
"""bench.py -- star-likelihood evals/sec of the MI355X-native BASE-9 log-posterior path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric: "star-likelihood evals/sec ... on synthetic 50k-star x 8-filter
clusters at 1 GPU, with 1/2/4/8-GPU walker-parallel throughput"; configs[2] sharded 8 ways):
50 000 stars x 8 filters, PARSEC-shaped synthetic pack, 8 walkers per GPU (weak scaling: 64
walkers at 8 GPUs).  One "step" is one adaptive-Metropolis step of every walker: propose ->
log-posterior of the rank's walkers -> accept/reject, all on the GPU (one HIP launch per step behind
the C ABI, b9_mcmc_run_block); every 100 steps the ranks exchange one RCCL all-gather of
per-walker rows and re-derive the pooled proposal covariance.  Star data and model tables are
resident in HBM before the timed region starts.

value     = n_stars x total walkers x K / max-over-ranks wall time   (whole job, all GPUs)
roofline  = the dominant kernel (k_mcmc_step, the fused sampler step: star likelihood of the step's
            proposal + the previous step's accept/reject + the next step's candidate isochrones): algorithmic bytes per launch / its mean launch
            duration, measured with HIP events on the launch stream inside this run
cpu_baseline = the CPU oracle ("port"; the reference itself is not mounted)
            timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
N_STARS, N_FILT, WALKERS_PER_GPU = 50000, 8, 8
MCMC_BLOCK = 100       # steps between adaptation points (= between all-gathers)
TIMING_EVERY = 25       # a HIP-event bracket opens at every 25th launch of the dominant kernel in the timed region and spans 8 launches


def cpu_baseline(pack_d, cl, truth, budget_s: float = 14.0, eng=None):
    """Time the CPU oracle on a bounded sample of the same workload (rank 0 only): all host cores
    (OpenMP over stars, as the reference's thread pool [RECALL]) and one thread.  With `eng`, the same
    leg also reports BASELINE.json's second figure, |delta logPost| of the HIP path against that CPU
    path, over 128 random in-grid parameter rows on the full 50k-star cluster."""
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
        import numpy as np
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse ranks on one GPU)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="diagnostic: skip the HIP-event bracketing of k_mcmc_step")
    args = ap.parse_args()

    import torch
    from base_amd import abi, engine, mcmc, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    local_rank = local_rank % max(1, torch.cuda.device_count())      # rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("B9_FORCE_DIST") == "1"     # the latter: 1-rank rehearsal of the collectives
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(args.backend)

    # ---- synthetic inputs (identical on every rank: fixed seeds) -----------------------------
    pack_d = synth.make_pack("parsec", N_FILT)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, N_STARS, seed=9003, truth=truth)
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors, options = synth.default_priors(pack_d, truth), abi.make_options()
    eng = engine.Engine(pack, stars, priors, options, device=local_rank)
    n_walkers = WALKERS_PER_GPU * world
    start = synth.walker_params(truth, n_walkers, seed=42, scale=0.02)
    gather = mcmc.torch_all_gather("cuda" if args.backend == "nccl" else None) if use_dist else None
    block = MCMC_BLOCK
    sampler = mcmc.WalkerSampler(start, mcmc.DeviceBlockRunner(eng, record=True), rank, world, gather,
                                 seed=2024, block=block)
    sampler.initialise(eng.logpost)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    sampler.run(args.warmup)
    acc0 = sampler.accepted
    eng.enable_timing(0 if args.no_kernel_timing else TIMING_EVERY)     # also pre-creates the HIP-event pool
    eng.kernel_time_ms(reset=True)
    barrier()
    t0 = time.perf_counter()
    sampler.run(args.steps)             # exactly K steps, in device-resident blocks of <= MCMC_BLOCK
    barrier()
    dt = time.perf_counter() - t0
    k_ms, k_n = eng.kernel_time_ms(reset=True)
    eng.enable_timing(0)
    bracket_ms = eng.calibrate_timing()     # event bracket around an empty kernel, same stream

    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        evals = float(N_STARS) * n_walkers * args.steps
        bytes_eval = eng.bytes_per_star_eval()
        bytes_launch = float(bytes_eval) * N_STARS * WALKERS_PER_GPU
        # The HIP-event bracket includes one dispatch boundary, so it reads ~3 us above rocprofv3's
        # kernel-only average of the same command (profiles/); it is used as is (conservative).  The
        # same bracket around an EMPTY kernel is reported for scale, never subtracted.
        k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
        achieved = bytes_launch / k_avg_s / 1e9 if k_n > 0 else 0.0
        # HBM traffic per launch of the dominant kernel: from the committed PMC passes of this same
        # command (profiles/<round>_summary.json, tools/profile_round.sh); null when absent
        traffic = None
        for tag in ("r01",):
            pth = os.path.join(ROOT, "profiles", f"{tag}_summary.json")
            if os.path.exists(pth):
                pm = json.load(open(pth)).get("pmc", {})
                for kname, c in pm.items():
                    if kname.startswith("k_mcmc_step") and "hbm_bytes_per_launch" in c:
                        traffic = c["hbm_bytes_per_launch"]
        out = {
            "metric": "star-likelihood evals/sec", "value": evals / dt, "unit": "star-likelihood evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "mcmc_steps_per_s": args.steps / dt,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3: 50k-star x 8-filter synthetic cluster, PARSEC-shaped synthetic pack "
                                   "(10 FeH x 60 ages x 400 EEPs), given-mass mode, 8 walkers per GPU",
                       "n_stars": N_STARS, "n_filters": N_FILT, "walkers_per_gpu": WALKERS_PER_GPU,
                       "walkers_total": n_walkers, "parallelism": f"walkers{world}",
                       "mcmc_block": block,
                       "collective": "one all_gather of [logpost, position, moments] rows per 100-step block" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_mcmc_step", "launches_timed": k_n, "timed_every": TIMING_EVERY, "launches_per_bracket": 8,
                         "avg_launch_us": 1e6 * k_avg_s, "empty_kernel_bracket_us": 1e3 * bracket_ms,
                         "algorithmic_bytes_per_launch": bytes_launch, "bytes_per_star_eval": bytes_eval},
            "accept_rate": (sampler.accepted - acc0) / float(WALKERS_PER_GPU * args.steps),
            "parity": "vs this repo's CPU oracle (BASE-9 parity unpinned: reference source not mounted)",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pack_d, cl, truth, eng=eng)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if world == 1:
            out["marginalised_mode"] = marginalised_leg(pack, stars, priors, truth, local_rank)
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
