#!/usr/bin/env python3
"""Times the marginalised mode (k_marg_table + k_star_marg [+ k_star_marg_wd]) through b9_logpost, any instance:
    time_marg.py <n_stars> <K> <Q> <walkers> [--filters F] [--pops P] [--wd FRAC] [--sample] [--cpu]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, synth
a = sys.argv[1:]
n_stars, K, Q, W = int(a[0]), int(a[1]), int(a[2]), int(a[3])
opt_i = lambda name, d: int(a[a.index(name) + 1]) if name in a else d
nf, npops = opt_i("--filters", 8), opt_i("--pops", 1)
wd = float(a[a.index("--wd") + 1]) if "--wd" in a else 0.0
pack_d = synth.make_pack("parsec", nf, n_y=3 if npops == 2 else 1); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, n_stars, seed=9003, truth=truth, n_pops=npops, wd_frac=wd)
pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
opt = abi.make_options(abi.MODE_MARGINALISED, npops, K, Q)
eng = engine.Engine(pack, stars, synth.default_priors(pack_d, truth, npops), opt)
params = synth.walker_params(truth, W, seed=42, scale=0.05, n_pops=npops)
call = (lambda: eng.sample_mass(params, seed=3)) if "--sample" in a else (lambda: eng.logpost(params))
call()
t0 = time.perf_counter(); reps = 10
for _ in range(reps): call()
dt = (time.perf_counter() - t0) / reps
nodes = (eng.max_eep() - 1) * K * Q
print(f"GPU marg{' (sampleMass draws)' if '--sample' in a else ''}: {n_stars} stars x {nf} filters x {W} walkers, {npops} pop, {wd:.0%} WD, K={K} Q={Q} "
      f"({nodes} nodes/star): {dt*1e3:.2f} ms/call, {n_stars*W/dt:.3e} star-evals/s, {n_stars*W*nodes/dt:.3e} node-evals/s")
if "--cpu" in a:
    import oracle
    sub = {k: (np.asarray(v)[:50] if k in ("obs","sigma","mass1","mass_ratio","clust_prior","stage","wd_type") else v) for k, v in cl.items()}
    orc = oracle.Oracle(pack, abi.make_stars(sub), synth.default_priors(pack_d, truth, npops), opt)
    t0 = time.perf_counter(); orc.logpost(params[:1]); dt = time.perf_counter() - t0
    print(f"CPU oracle marg: 50 stars x 1 walker: {dt:.2f} s -> {50/dt:.3e} star-evals/s")
