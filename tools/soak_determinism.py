#!/usr/bin/env python3
"""Soak: the device-resident sampler run twice from the same state gives bit-identical chains (fixed summation
orders, counter-based RNG, no data-path atomics), over many blocks -- on the bench shape (one-step launch) and, with a
shape argument, on the single-chain shapes (tree launch, WD stars, two populations) against the host twin's first block too.
    soak_determinism.py [n_blocks] [C2|C1|C3|W2|P2|C4|C4W] [--marg K Q [--calls N]]
--marg K Q: the marginalised mode (k_marg_step: fused step, split and unsplit instances by catalogue size) -- besides the two
sampler runs, N (default 200) repeated b9_logpost(perstar=True) calls of one set of rows must return the same bits for every
star every time (the pruning reference of k_star_marg is a function of the data only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, mcmc, synth
args = sys.argv[1:]
MARG = None
if "--marg" in args:
    k = args.index("--marg")
    MARG = (int(args[k + 1]), int(args[k + 2]))
    del args[k:k + 3]
n_calls = 200
if "--calls" in args:
    k = args.index("--calls")
    n_calls = int(args[k + 1])
    del args[k:k + 2]
n_blocks = int(args[0]) if len(args) > 0 else 200
shape = args[1] if len(args) > 1 else "C2"
pk, nf, ns, wd, ny, npops, W = {"C2": ("parsec", 8, 50000, 0.0, 1, 1, 8), "C1": ("dsed", 8, 10000, 0.0, 1, 1, 1), "C3": ("parsec", 8, 20000, 0.05, 1, 1, 1),
                                 "W2": ("parsec", 8, 50000, 0.01, 1, 1, 2), "P2": ("parsec", 5, 6000, 0.03, 3, 2, 1),
                                 "C4": ("parsec", 8, 30000, 0.0, 3, 2, 8), "C4W": ("parsec", 8, 30000, 0.02, 3, 2, 8)}[shape]
pack_d = synth.make_pack(pk, nf, n_y=ny); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, ns, seed=9003, truth=truth, wd_frac=wd, n_pops=npops)
opt = abi.make_options(abi.MODE_MARGINALISED, npops, MARG[0], MARG[1]) if MARG else abi.make_options(n_pops=npops)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth, npops), opt)
free = np.array(mcmc.DEFAULT_FREE if npops == 1 else mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA))
chol = np.diag([2e-5, 1e-4, 4e-5, 4e-5] + ([3e-5, 3e-5, 2e-4] if npops == 2 else [])) * (1.0 if shape == "C2" else 3.0)
start = synth.walker_params(truth, W, seed=42, scale=0.02, n_pops=npops)
lp0 = eng.logpost(start)
mode = f", marginalised {MARG[0]} x {MARG[1]}" if MARG else ""
print(f"{shape}: {ns} stars, {W} walker(s), {npops} pop(s){mode}, {eng.step_depth(W)} step(s) per launch")
ok_calls = True
if MARG:
    ref_lp, ref_ps = eng.logpost(start, perstar=True)
    t0 = time.perf_counter()
    bad = 0
    for i in range(n_calls):
        lp, ps = eng.logpost(start, perstar=True)
        bad += int(not (np.array_equal(lp, ref_lp) and np.array_equal(ps, ref_ps, equal_nan=True)))
    ok_calls = bad == 0
    print(f"{n_calls} repeated b9_logpost(perstar) calls in {time.perf_counter()-t0:.2f} s: {bad} differ from the first in any bit ({ref_ps.size} per-star values each)")
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
sys.exit(0 if same and twin and ok_calls else 1)
