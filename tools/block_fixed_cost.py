#!/usr/bin/env python3
"""Fixed cost of one sampler block at the C ABI: wall time of b9_mcmc_run_block (synchronous, CONTINUE off / on through
the pipelined sampler) for several block lengths, and the least-squares intercept (us per block) and slope (us per step)."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(45, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, hostlib, mcmc, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
start = synth.walker_params(truth, 8, seed=42, scale=0.02)
lp = eng.logpost(start)
free = np.array(mcmc.DEFAULT_FREE, dtype=np.int32); chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
ids = np.arange(8, dtype=np.int32)
eng.mcmc_run_block(start, lp, ids, free, chol, 1, 0, 300, record=False)
res = {}
for n in (5, 10, 20, 40, 80, 160):
    best = 1e9
    for rep in range(7):
        time.sleep(0.002)                      # the GPU idles between measurements, as it does before bench.py's timed region
        t0 = time.perf_counter()
        eng.mcmc_run_block(start, lp, ids, free, chol, 1, 0, n, record=False)
        best = min(best, time.perf_counter() - t0)
    res[n] = best * 1e6
    print(f"synchronous block of {n:4d} steps: {best * 1e6:8.1f} us  ({best * 1e6 / n:6.2f} us/step)")
ns = np.array(list(res)); ts = np.array([res[n] for n in ns])
slope, icpt = np.polyfit(ns, ts, 1)
print(f"fit: {icpt:.1f} us per block + {slope:.2f} us per step")
print("creating sampler", flush=True)
s = hostlib.HostSampler(8, free, [mcmc.DEFAULT_STEP[int(k)] for k in free], hostlib.Exchange.local(), seed=5, block=100, engine=eng)
s.initialise(start); print("initialised", flush=True); s.run(300); print("ran", flush=True)
for n in (20, 100):
    best = 1e9
    for rep in range(7):
        time.sleep(0.002)
        t0 = time.perf_counter(); s.run(n); best = min(best, time.perf_counter() - t0)
    print(f"C++ sampler run({n}): {best * 1e6:8.1f} us  ({best * 1e6 / n:6.2f} us/step)")
