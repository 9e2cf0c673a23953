#!/usr/bin/env python3
"""Soak: the device-resident sampler run twice from the same state gives bit-identical chains (fixed summation
orders, counter-based RNG, no data-path atomics), over many blocks on the bench shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, mcmc, synth
n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
free = np.array(mcmc.DEFAULT_FREE); chol = np.diag([2e-5, 1e-4, 4e-5, 4e-5])
start = synth.walker_params(truth, 8, seed=42, scale=0.02)
lp0 = eng.logpost(start)
outs = []
for rep in range(2):
    p, lp, acc = start.copy(), lp0.copy(), 0
    t0 = time.perf_counter()
    for b in range(n_blocks):
        p, lp, s, l, a = eng.mcmc_run_block(p, lp, np.arange(8), free, chol, 99, b * 100, 100)
        acc += a
    outs.append((p, lp, acc, s, l))
    print(f"run {rep}: {n_blocks*100} steps in {time.perf_counter()-t0:.2f} s, accepted {acc}")
same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]) if isinstance(a, np.ndarray)) and outs[0][2] == outs[1][2]
print("bit-identical:", same)
sys.exit(0 if same else 1)
