#!/usr/bin/env python3
"""Latency of one host-driven b9_logpost call (what a per-step binding of the reference pays): the parameter rows in the
first launch's kernel arguments, derive + stars + finalize (marginalised mode: derive + node table + stars [+ merge] + finalize),
the log-posteriors through mapped host memory behind a polled completion word.   usage: time_logpost.py [--marg K Q]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, synth
args = sys.argv[1:]
MARG = (int(args[args.index("--marg") + 1]), int(args[args.index("--marg") + 2])) if "--marg" in args else None
for label, name, n_stars, W in (("C1", "dsed", 10000, 1), ("C2 catalogue, one row", "parsec", 50000, 1), ("C2", "parsec", 50000, 8)):
    pack_d = synth.make_pack(name, 8); truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, n_stars, seed=9001, truth=truth)
    opt = abi.make_options(abi.MODE_MARGINALISED, 1, MARG[0], MARG[1]) if MARG else abi.make_options()
    eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), opt)
    rows = synth.walker_params(truth, W, seed=1, scale=0.02)
    for _ in range(50): eng.logpost(rows)
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n): eng.logpost(rows)
    dt = (time.perf_counter() - t0) / n
    mode = f"marginalised {MARG[0]} x {MARG[1]}" if MARG else "given-mass"
    print(f"{label}: {name} {n_stars} stars x {W} row(s), {mode}: {dt*1e6:.1f} us per b9_logpost call ({n_stars*W/dt:.3e} star-evals/s)")
    eng.close()
