#!/usr/bin/env python3
"""Diagnostic: wave start/end timeline (entry and exit stamps only, so the stamps barely perturb the
kernel) of k_mcmc_step's or k_star_like's hot waves.  Needs build/variants/lib_stamps01.so
(-DB9_STAMPS -DB9_STAMP_MASK=0x101).   usage: stamps_life.py step|like"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["B9_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build/variants/lib_stamps01.so")
import numpy as np
from base_amd import abi, engine, synth
which = sys.argv[1] if len(sys.argv) > 1 else "step"
n_stars, n_walkers = 50000, 8
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, n_stars, seed=9003, truth=truth)
pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
eng = engine.Engine(pack, stars, synth.default_priors(pack_d, truth), abi.make_options())
params = synth.walker_params(truth, n_walkers, seed=42, scale=0.05)
lp = eng.logpost(params)
free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS], dtype=np.int32)
chol = np.diag([2e-4, 2e-3, 5e-4, 5e-4])
eng.lib.b9_debug_clear_stamps.restype = C.c_int
if which == "step":
    eng.mcmc_run_block(params, lp, np.arange(n_walkers, dtype=np.int32), free, chol, 7, 0, 40, record=False)
    assert eng.lib.b9_debug_clear_stamps() == 0
    eng.mcmc_run_block(params, lp, np.arange(n_walkers, dtype=np.int32), free, chol, 7, 0, 41, record=False)
else:
    for _ in range(5): eng.logpost(params)
    assert eng.lib.b9_debug_clear_stamps() == 0
    eng.logpost(params)
nw = 8192
buf = np.zeros((nw, 12), dtype=np.uint64)
eng.lib.b9_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert eng.lib.b9_debug_read_stamps(buf.ctypes.data, nw) == 0
t = buf.astype(np.int64)
t = t[(t[:, 8] > 0) & (t[:, 0] > 0)]
t0 = t[:, 0].min()
life = t[:, 8] - t[:, 0]
print(f"{which}: hot waves {len(t)}  span {t[:,8].max()-t0} cycles")
print("  lifetime mean %.0f p5 %.0f p50 %.0f p95 %.0f" % (life.mean(), *np.percentile(life, [5, 50, 95])))
if which == "step" and (t[:, 3] > 0).all():
    for a, b, nm in ((0, 1, "entry -> loads issued (+wait)"), (1, 2, "decision"), (2, 3, "LDS fill + barrier"), (3, 8, "tiles + reduce")):
        x = t[:, b] - t[:, a]; print(f"    {nm:30s} mean {x.mean():8.0f} p50 {np.median(x):8.0f} p95 {np.percentile(x,95):8.0f}")
print("  start offsets p5/p25/p50/p75/p95/max", np.percentile(t[:, 0] - t0, [5, 25, 50, 75, 95, 100]).astype(int))
print("  end   offsets p5/p25/p50/p75/p95/max", np.percentile(t[:, 8] - t0, [5, 25, 50, 75, 95, 100]).astype(int))
early = t[:, 0] - t0 < np.percentile(t[:, 0] - t0, 40)
print("  lifetime of first-round waves mean %.0f, later waves mean %.0f" % (life[early].mean(), life[~early].mean()))
