#!/usr/bin/env python3
"""Where a sampler block's wall time goes on the bench shape: the C call b9_mcmc_run_block vs the Python
driver around it (adaptation, row packing, gather).  usage: block_overhead.py [block=100] [blocks=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, synth, mcmc
block = int(sys.argv[1]) if len(sys.argv) > 1 else 100
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
start = synth.walker_params(truth, 8, seed=42, scale=0.05)
runner = mcmc.DeviceBlockRunner(eng, record=True)
t_c = [0.0]
def wrap(f):
    def timed(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); t_c[0] += time.perf_counter() - t0; return r
    return timed
runner.run, runner.submit, runner.collect = wrap(runner.run), wrap(runner.submit), wrap(runner.collect)   # C calls (enqueue + wait)
s = mcmc.WalkerSampler(start, runner, block=block)
s.initialise(eng.logpost)
s.run(10 * block)
t_c[0] = 0.0
t0 = time.perf_counter(); s.run(blocks * block); tot = time.perf_counter() - t0
n = blocks * block
print(f"block {block}: total {tot/n*1e6:.2f} us/step; inside the C calls (enqueue + wait) {t_c[0]/n*1e6:.2f} us/step; python around them {(tot-t_c[0])/blocks*1e6:.0f} us/block")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); s.run(10 * block); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
