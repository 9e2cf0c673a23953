#!/usr/bin/env python3
"""Two-population config (BASELINE configs[4] shape at one GPU's share): device-resident MCMC timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, mcmc, synth
pack_d = synth.make_pack("parsec", 8, n_y=3); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 30000, seed=9005, truth=truth, n_pops=2)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth, 2), abi.make_options(n_pops=2))
start = synth.walker_params(truth, 8, seed=7, n_pops=2, scale=0.02)
lp = eng.logpost(start)
free = np.array(mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA)); chol = np.diag([1e-5] * 7)
eng.mcmc_run_block(start, lp, np.arange(8), free, chol, 1, 0, 100)
t0 = time.perf_counter(); eng.mcmc_run_block(start, lp, np.arange(8), free, chol, 1, 0, 1000); dt = time.perf_counter() - t0
print(f"two-pop 30k x 8 walkers: {1e6*dt/1000:.2f} us/step, {30000*8*1000/dt:.3e} star-evals/s")
