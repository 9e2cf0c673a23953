#!/usr/bin/env python3
"""Runs the five BASELINE.json configurations at ONE GPU's share: parity vs the CPU oracle (full size: every star) +
device-resident MCMC throughput through the C++ driver.  Output: one JSON line per config (+ a table)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from base_amd import abi, engine, hostlib, mcmc, synth

CONFIGS = [(k,) + v + (False,) for k, v in synth.BASELINE_CONFIGS.items()]
# C3 again with RAGGED WD cooling tracks (every track its own age axis, as real cooling models have: the ABI-2 search path
# instead of the rectangular table's shared-axis fast path)
CONFIGS.append(("C3r",) + synth.BASELINE_CONFIGS["C3"][:-1] + ("C3 with ragged WD cooling tracks (per-track age axes)", True))
SUBKEYS = ("obs", "sigma", "mass1", "mass_ratio", "clust_prior", "stage", "wd_type")
rows = []
for name, pk, nf, ns, wd, ny, npops, W, note, ragged in CONFIGS:
    pack_d = synth.make_pack(pk, nf, n_y=ny, wd_ragged=ragged)
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
    r = dict(config=name, note=note, steps_per_launch=eng.step_depth(W), n_stars=ns, n_filt=nf, walkers=W, n_pops=npops, wd_stars=int((cl["stage"] == 3).sum()),
             max_rel_err_vs_oracle=err, oracle_subset=len(idx), mcmc_steps_per_s=2000 / dt, us_per_step=1e6 * dt / 2000,
             star_evals_per_s=2000 * W * ns / dt, accept_rate=accepted / (2000.0 * W))
    rows.append(r)
    print(json.dumps(r), flush=True)
    eng.close()
print("Five BASELINE.json configurations at one GPU's share (tools/config_sweep.py; C++ driver b9h::WalkerSampler, device-resident 100-step blocks;\n"
      "steps/launch 1 = the one-step fused launch k_mcmc_step, 2-3 = the tree-speculative launch k_mcmc_tree)\n")
print("| config | stars x filters | walkers | steps/launch | us/step | star-evals/s | max rel err vs oracle (all stars) | accept |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['config']} | {r['n_stars']} x {r['n_filt']}{' (2 pops)' if r['n_pops']==2 else ''}{' (%d WD)' % r['wd_stars'] if r['wd_stars'] else ''} | {r['walkers']} | {r['steps_per_launch']} | "
          f"{r['us_per_step']:.1f} | {r['star_evals_per_s']:.3e} | {r['max_rel_err_vs_oracle']:.1e} | {r['accept_rate']:.2f} |")
