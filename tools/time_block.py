#!/usr/bin/env python3
"""Where does a sampler block spend its time?  C call (b9_mcmc_run_block) vs Python bookkeeping."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, mcmc, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
eng = engine.Engine(pack, stars, synth.default_priors(pack_d, truth), abi.make_options())
start = synth.walker_params(truth, 8, seed=42, scale=0.02)
lp = eng.logpost(start)
free = np.array(mcmc.DEFAULT_FREE); chol = np.diag([1e-5, 2e-5, 1e-5, 1e-5])
ids = np.arange(8)
for block in (50, 200, 1000):
    eng.mcmc_run_block(start, lp, ids, free, chol, 1, 0, block)
    t0 = time.perf_counter(); n = 0
    while n < 2000:
        eng.mcmc_run_block(start, lp, ids, free, chol, 1, n, block); n += block
    dt = time.perf_counter() - t0
    print(f"C call only, block={block}: {1e6*dt/n:.2f} us/step")
for block in (50, 200):
    s = mcmc.WalkerSampler(start, mcmc.DeviceBlockRunner(eng), block=block, seed=5)
    s.initialise(eng.logpost); s.run(2 * block)
    t0 = time.perf_counter(); s.run(2000); dt = time.perf_counter() - t0
    print(f"WalkerSampler, block={block}: {1e6*dt/2000:.2f} us/step")
