#!/usr/bin/env python3
"""Throughput of the device-resident sampler vs walkers per GPU (50k stars x 8 filters)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, mcmc, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
free = np.array(mcmc.DEFAULT_FREE); chol = np.diag([1e-5, 2e-5, 1e-5, 1e-5])
_w = synth.walker_params(truth, 8, seed=1, scale=0.02)           # discard run: clocks / first-use effects
eng.mcmc_run_block(_w, eng.logpost(_w), np.arange(8), free, chol, 1, 0, 1500)
print("| walkers | us/step | star-evals/s | algorithmic GB/s (152 B per star-eval, SURVEY 8d; an L2-served rate below ~16 walkers) |\n|---|---|---|---|")
for W in (1, 2, 4, 8, 16, 32, 64, 128):
    start = synth.walker_params(truth, W, seed=42, scale=0.02)
    lp = eng.logpost(start)
    eng.mcmc_run_block(start, lp, np.arange(W), free, chol, 1, 0, 100)
    n = 600 if W <= 16 else 200
    t0 = time.perf_counter(); eng.mcmc_run_block(start, lp, np.arange(W), free, chol, 1, 0, n); dt = time.perf_counter() - t0
    print(f"| {W} | {1e6*dt/n:.1f} | {50000*W*n/dt:.3e} | {152*50000*W*n/dt/1e9:.0f} |", flush=True)
