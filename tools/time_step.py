#!/usr/bin/env python3
"""C-call-level time of the device-resident sampler step (one 1000-step block after a 200-step warm-up) for the
BASELINE shapes at one GPU's share.   usage: [B9_HIP_LIB=...] time_step.py C2 [C4 ...] [--walkers N] [--marg K Q] [--tune field=value ...]
(--marg: the marginalised mode with K sub-steps per EEP interval x Q mass ratios: two launches per step + the node table)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, mcmc, synth
SHAPES = {"C0": ("girardi", 3, 200, 0.0, 1, 1, 1), "C1": ("dsed", 8, 10000, 0.0, 1, 1, 1), "C2": ("parsec", 8, 50000, 0.0, 1, 1, 8),
          "C3": ("parsec", 8, 20000, 0.05, 1, 1, 1), "C4": ("parsec", 8, 30000, 0.0, 3, 2, 8),
          "F16": ("parsec", 16, 50000, 0.0, 1, 1, 8), "F16P2": ("parsec", 16, 30000, 0.0, 3, 2, 8), "F4": ("parsec", 4, 50000, 0.0, 1, 1, 8)}
args = sys.argv[1:]
Wo = int(args[args.index("--walkers") + 1]) if "--walkers" in args else None
names = [a for a in args if a in SHAPES] or ["C2"]
MARG = (int(args[args.index("--marg") + 1]), int(args[args.index("--marg") + 2])) if "--marg" in args else None
TUNE = {a.split("=")[0]: int(a.split("=")[1]) for i, a in enumerate(args) if i and args[i - 1] == "--tune"}      # b9_tuning fields (marg_piece_units=8)
tag = os.path.basename(os.environ.get("B9_HIP_LIB", "libbase9hip.so"))
for name in names:
    pk, nf, ns, wd, ny, npops, W = SHAPES[name]
    W = Wo or W
    pack_d = synth.make_pack(pk, nf, n_y=ny); truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, ns, seed=9001 + (int(name[1]) if name[1].isdigit() else 7), truth=truth, wd_frac=wd, n_pops=npops)
    eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth, npops), abi.make_options(abi.MODE_MARGINALISED, npops, MARG[0], MARG[1]) if MARG else abi.make_options(n_pops=npops))
    if TUNE:
        eng.update_tuning(**TUNE)
    free = np.array(mcmc.DEFAULT_FREE if npops == 1 else mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA), dtype=np.int32)
    start = synth.walker_params(truth, W, seed=7, n_pops=npops, scale=0.02)
    lp = eng.logpost(start)
    chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
    ids = np.arange(W, dtype=np.int32)
    eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 200, record=False)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 1000, record=False)
        best = min(best, time.perf_counter() - t0)
    print(f"{tag:28s} {name}{' marg %dx%d' % MARG if MARG else ''}{' ' + str(TUNE) if TUNE else ''}: {ns} x {nf} x {W} walkers, {npops} pop: {1e3 * best:.2f} us/step  {ns * W * 1000 / best:.3e} star-evals/s", flush=True)
    eng.close()
