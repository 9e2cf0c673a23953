#!/usr/bin/env python3
"""Probe: does running the marginalised sampler's walkers as TWO independent half-ensembles on two streams (two contexts,
two host threads) hide the small per-step launches (k_derive_iso, k_marg_table) of one half behind the other's k_star_marg?
Prints steps/s of 8 walkers in one context against 4 + 4 in two."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from base_amd import abi, engine, mcmc, synth
N, K, Q, W, STEPS = 50000, 4, 4, 8, 400
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, N, seed=9003, truth=truth)
pack, stars, priors = abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth)
opt = abi.make_options(abi.MODE_MARGINALISED, 1, K, Q)
free = np.array(mcmc.DEFAULT_FREE, dtype=np.int32)
chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
start = synth.walker_params(truth, W, seed=7, scale=0.02)

def run(eng, rows, ids, out, key):
    lp = eng.logpost(rows)
    eng.mcmc_run_block(rows, lp, ids, free, chol, 7, 0, 50, record=False)
    t0 = time.perf_counter()
    eng.mcmc_run_block(rows, lp, ids, free, chol, 7, 0, STEPS, record=False)
    out[key] = time.perf_counter() - t0

res = {}
e8 = engine.Engine(pack, stars, priors, opt)
run(e8, start, np.arange(W, dtype=np.int32), res, "one")
print(f"one context, {W} walkers: {1e6 * res['one'] / STEPS:.1f} us per step")
ea, eb = engine.Engine(pack, stars, priors, opt), engine.Engine(pack, stars, priors, opt)
bar = threading.Barrier(2)
def worker(eng, lo, key):
    rows, ids = start[lo:lo + W // 2].copy(), np.arange(lo, lo + W // 2, dtype=np.int32)
    lp = eng.logpost(rows)
    eng.mcmc_run_block(rows, lp, ids, free, chol, 7, 0, 50, record=False)
    bar.wait()
    t0 = time.perf_counter()
    eng.mcmc_run_block(rows, lp, ids, free, chol, 7, 0, STEPS, record=False)
    res[key] = time.perf_counter() - t0
ts = [threading.Thread(target=worker, args=(ea, 0, "a")), threading.Thread(target=worker, args=(eb, W // 2, "b"))]
[t.start() for t in ts]; [t.join() for t in ts]
print(f"two contexts, {W // 2} + {W // 2} walkers: {1e6 * max(res['a'], res['b']) / STEPS:.1f} us per step of all {W} (halves: {1e6 * res['a'] / STEPS:.1f}, {1e6 * res['b'] / STEPS:.1f})")
