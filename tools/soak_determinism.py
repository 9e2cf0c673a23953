#!/usr/bin/env python3
"""Soak: the device-resident sampler run twice from the same state gives bit-identical chains (fixed summation
orders, counter-based RNG, no data-path atomics), over many blocks -- on the bench shape (one-step launch) and, with a
shape argument, on the single-chain shapes (tree launch, WD stars, two populations) against the host twin's first block too.
    soak_determinism.py [n_blocks] [C2|C1|C3|W2|P2|C4|C4W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, mcmc, synth
n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shape = sys.argv[2] if len(sys.argv) > 2 else "C2"
pk, nf, ns, wd, ny, npops, W = {"C2": ("parsec", 8, 50000, 0.0, 1, 1, 8), "C1": ("dsed", 8, 10000, 0.0, 1, 1, 1), "C3": ("parsec", 8, 20000, 0.05, 1, 1, 1),
                                 "W2": ("parsec", 8, 50000, 0.01, 1, 1, 2), "P2": ("parsec", 5, 6000, 0.03, 3, 2, 1),
                                 "C4": ("parsec", 8, 30000, 0.0, 3, 2, 8), "C4W": ("parsec", 8, 30000, 0.02, 3, 2, 8)}[shape]
pack_d = synth.make_pack(pk, nf, n_y=ny); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, ns, seed=9003, truth=truth, wd_frac=wd, n_pops=npops)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth, npops), abi.make_options(n_pops=npops))
free = np.array(mcmc.DEFAULT_FREE if npops == 1 else mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA))
chol = np.diag([2e-5, 1e-4, 4e-5, 4e-5] + ([3e-5, 3e-5, 2e-4] if npops == 2 else [])) * (1.0 if shape == "C2" else 3.0)
start = synth.walker_params(truth, W, seed=42, scale=0.02, n_pops=npops)
lp0 = eng.logpost(start)
print(f"{shape}: {ns} stars, {W} walker(s), {npops} pop(s), {eng.step_depth(W)} step(s) per launch")
outs = []
for rep in range(2):
    p, lp, acc = start.copy(), lp0.copy(), 0
    t0 = time.perf_counter()
    for b in range(n_blocks):
        p, lp, s, l, a = eng.mcmc_run_block(p, lp, np.arange(W), free, chol, 99, b * 100, 100)
        acc += a
    outs.append((p, lp, acc, s, l))
    print(f"run {rep}: {n_blocks*100} steps in {time.perf_counter()-t0:.2f} s, accepted {acc}")
same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]) if isinstance(a, np.ndarray)) and outs[0][2] == outs[1][2]
print("bit-identical:", same)
host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, np.arange(W), free, chol, 99, 0, 100)
dev = eng.mcmc_run_block(start, lp0, np.arange(W), free, chol, 99, 0, 100)
twin = dev[4] == host[4] and np.allclose(dev[3], host[3], rtol=1e-10, atol=0) and np.allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
print("first block equals the host twin's:", twin, f"(accepted {dev[4]})")
sys.exit(0 if same and twin else 1)
