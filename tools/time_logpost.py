#!/usr/bin/env python3
"""Latency of one host-driven b9_logpost call (what a per-step binding of the reference pays):
upload of the parameter rows + derive + stars + finalize + download."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, synth
for name, n_stars, W in (("dsed", 10000, 1), ("parsec", 50000, 1), ("parsec", 50000, 8)):
    pack_d = synth.make_pack(name, 8); truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, n_stars, seed=9001, truth=truth)
    eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
    rows = synth.walker_params(truth, W, seed=1, scale=0.02)
    for _ in range(50): eng.logpost(rows)
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n): eng.logpost(rows)
    dt = (time.perf_counter() - t0) / n
    print(f"{name} {n_stars} stars x {W} walker(s): {dt*1e6:.1f} us per b9_logpost call ({n_stars*W/dt:.3e} star-evals/s)")
    eng.close()
