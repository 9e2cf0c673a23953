#!/usr/bin/env python3
"""Times the marginalised mode (k_star_marg) and the CPU oracle on a small sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, synth
n_stars, K, Q, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, n_stars, seed=9003, truth=truth)
pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
opt = abi.make_options(abi.MODE_MARGINALISED, 1, K, Q)
eng = engine.Engine(pack, stars, synth.default_priors(pack_d, truth), opt)
params = synth.walker_params(truth, W, seed=42, scale=0.05)
eng.logpost(params)
t0 = time.perf_counter(); reps = 10
for _ in range(reps): lp = eng.logpost(params)
dt = (time.perf_counter() - t0) / reps
nodes = 399 * K * Q
print(f"GPU marg: {n_stars} stars x {W} walkers, K={K} Q={Q} ({nodes} nodes/star): {dt*1e3:.2f} ms/call, "
      f"{n_stars*W/dt:.3e} star-evals/s, {n_stars*W*nodes/dt:.3e} node-evals/s")
if "--cpu" in sys.argv:
    import oracle
    sub = {k: (np.asarray(v)[:50] if k in ("obs","sigma","mass1","mass_ratio","clust_prior","stage","wd_type") else v) for k, v in cl.items()}
    orc = oracle.Oracle(pack, abi.make_stars(sub), synth.default_priors(pack_d, truth), opt)
    t0 = time.perf_counter(); orc.logpost(params[:1]); dt = time.perf_counter() - t0
    print(f"CPU oracle marg: 50 stars x 1 walker: {dt:.2f} s -> {50/dt:.3e} star-evals/s")
