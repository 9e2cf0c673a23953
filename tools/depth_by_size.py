"""Sampler step time of one context by catalogue size and walker count, with the launch depth the plan chose.

usage: [B9_TREE_DEPTH=1|2|3] python tools/depth_by_size.py N_STARS N_WALKERS   (GPU box)
"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from base_amd import abi, engine, mcmc, synth
ns, W = int(sys.argv[1]), int(sys.argv[2])
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, ns, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
free = np.array(mcmc.DEFAULT_FREE, dtype=np.int32)
start = synth.walker_params(truth, W, seed=7, scale=0.02); lp = eng.logpost(start)
chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
ids = np.arange(W, dtype=np.int32)
eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 100, record=False)
t0 = time.perf_counter(); eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 300, record=False); dt = time.perf_counter() - t0
print(f"{ns} stars x {W} walkers, B9_TREE_DEPTH={os.environ.get('B9_TREE_DEPTH','auto')}: depth {eng.step_depth(W)}, {1e6*dt/300:.2f} us/step")
