#!/usr/bin/env python3
"""Runs the five BASELINE.json configurations at ONE GPU's share: parity vs the CPU oracle (full size: every star) +
device-resident MCMC throughput through the C++ driver.  Output: one JSON line per config (+ a table)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from base_amd import abi, engine, hostlib, mcmc, synth

CONFIGS = [
    # name, pack, n_filt, n_stars, wd_frac, n_y, n_pops, walkers on this GPU, note
    ("C0", "girardi", 3, 200, 0.0, 1, 1, 1, "200-star, Girardi-shaped, 3 filters, 1 chain (the reference's CPU plumbing case)"),
    ("C1", "dsed", 8, 10000, 0.0, 1, 1, 1, "10k-star, DSED-shaped, 8 filters, 1 chain, 1 GPU"),
    ("C2", "parsec", 8, 50000, 0.0, 1, 1, 8, "50k-star, PARSEC-shaped, 8 filters, 64 walkers / 8 GPUs -> 8 per GPU (bench.py workload)"),
    ("C3", "parsec", 8, 20000, 0.05, 1, 1, 1, "20k-star mixed MS+WD (5% WD: Bergeron-like atmospheres + IFMR), 8 filters, 1 GPU"),
    ("C4", "parsec", 8, 30000, 0.0, 3, 2, 8, "two-population 30k-star, 8 filters, 32 walkers / 4 GPUs -> 8 per GPU"),
]
SUBKEYS = ("obs", "sigma", "mass1", "mass_ratio", "clust_prior", "stage", "wd_type")
rows = []
for name, pk, nf, ns, wd, ny, npops, W, note in CONFIGS:
    pack_d = synth.make_pack(pk, nf, n_y=ny)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, ns, seed=9001 + int(name[1]), truth=truth, wd_frac=wd, n_pops=npops)
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors, options = synth.default_priors(pack_d, truth, npops), abi.make_options(n_pops=npops)
    eng = engine.Engine(pack, stars, priors, options)
    params = synth.walker_params(truth, max(W, 2), seed=42, n_pops=npops, scale=0.3)
    lp, ps = eng.logpost(params, perstar=True)
    idx = np.arange(ns)                       # every star (the OpenMP oracle takes milliseconds)
    sub = {k: (np.asarray(v)[idx] if k in SUBKEYS else v) for k, v in cl.items()}
    want = oracle.Oracle(pack, abi.make_stars(sub), priors, options).logpost(params, perstar=True)[1]
    got = ps[:, idx]
    err = float(np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))))
    free = mcmc.DEFAULT_FREE if npops == 1 else mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA)
    start = synth.walker_params(truth, W, seed=7, n_pops=npops, scale=0.02)
    # the C++ driver (b9h::WalkerSampler): device-resident pipelined blocks, as bench.py and the CLI run it
    s = hostlib.HostSampler(W, free, [mcmc.DEFAULT_STEP[k] for k in free], hostlib.Exchange.local(), seed=11, block=100, engine=eng)
    s.initialise(start)
    s.run(500)
    a0 = s.state()["accepted_local"]
    t0 = time.perf_counter(); s.run(2000); dt = time.perf_counter() - t0
    accepted = s.state()["accepted_local"] - a0
    r = dict(config=name, note=note, n_stars=ns, n_filt=nf, walkers=W, n_pops=npops, wd_stars=int((cl["stage"] == 3).sum()),
             max_rel_err_vs_oracle=err, oracle_subset=len(idx), mcmc_steps_per_s=2000 / dt, us_per_step=1e6 * dt / 2000,
             star_evals_per_s=2000 * W * ns / dt, accept_rate=accepted / (2000.0 * W))
    rows.append(r)
    print(json.dumps(r), flush=True)
    eng.close()
print("Five BASELINE.json configurations at one GPU's share (tools/config_sweep.py; C++ driver, fused one-launch sampler step, 100-step blocks)\n")
print("| config | stars x filters | walkers | us/step | star-evals/s | max rel err vs oracle (all stars) | accept |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['config']} | {r['n_stars']} x {r['n_filt']}{' (2 pops)' if r['n_pops']==2 else ''}{' (%d WD)' % r['wd_stars'] if r['wd_stars'] else ''} | {r['walkers']} | "
          f"{r['us_per_step']:.1f} | {r['star_evals_per_s']:.3e} | {r['max_rel_err_vs_oracle']:.1e} | {r['accept_rate']:.2f} |")
