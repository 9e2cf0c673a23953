import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from base_amd import abi, engine, mcmc, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
free = np.array(mcmc.DEFAULT_FREE); chol = np.diag([1e-5, 2e-5, 1e-5, 1e-5])
for tpb in (1, 2, 3, 4, 6, 8):
  for parts in (2, 4):
    os.environ["B9_TILES_PER_BLOCK"] = str(tpb); os.environ["B9_DERIVE_PARTS"] = str(parts)
    eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
    out = []
    for W in (16, 24, 32, 64):
        start = synth.walker_params(truth, W, seed=42, scale=0.02)
        lp = eng.logpost(start)
        eng.mcmc_run_block(start, lp, np.arange(W), free, chol, 1, 0, 100)
        n = 200
        t0 = time.perf_counter(); eng.mcmc_run_block(start, lp, np.arange(W), free, chol, 1, 0, n); dt = time.perf_counter() - t0
        out.append(f"W{W}: {1e6*dt/n:.1f}")
    print(f"tpb {tpb} parts {parts}: " + "  ".join(out), flush=True)
    del eng
