"""Host enqueue time against GPU time of short sampler blocks (the driver times 20-step calls): per block the host spends
~4 us per launch; a 20-step block of the bench shape takes ~360 us = 20 x 15.1 us of step launches + 24 us of opening /
first derivation / closing launches + ~30 us of start-up and completion latency.   usage: python tools/time_enqueue.py  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from base_amd import abi, engine, mcmc, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
free = np.array(mcmc.DEFAULT_FREE, dtype=np.int32); W = 8
start = synth.walker_params(truth, W, seed=7, scale=0.02); lp = eng.logpost(start)
chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
ids = np.arange(W, dtype=np.int32)
eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 500, record=False)

for K in (20, 20, 20, 100, 100, 100, 1000, 1000):
    eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 200, record=False)
    t0 = time.perf_counter()
    h = eng.mcmc_submit(start, lp, ids, free, chol, 7, 0, K, record=False, row_origin=start[0, free])
    t1 = time.perf_counter()
    eng.mcmc_collect(h)
    t2 = time.perf_counter()
    print(f"K={K}: enqueue {1e6*(t1-t0):.1f} us ({1e6*(t1-t0)/(K+3):.2f} us per launch), wait {1e6*(t2-t1):.1f} us, total {1e6*(t2-t0):.1f} us = {1e6*(t2-t0)/K:.2f} us/step; kernels alone ~{K*15.1+24:.0f} us")
