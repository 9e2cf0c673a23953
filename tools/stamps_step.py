#!/usr/bin/env python3
"""Diagnostic: per-phase wave timeline of the hot role of k_mcmc_step (fused sampler step) from the
-DB9_STAMPS build (build/variants/lib_stamps.so).  The stamps of the block's LAST step launch remain."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["B9_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build/variants/lib_stamps.so")
import numpy as np
from base_amd import abi, engine, synth
n_stars, n_walkers = 50000, 8
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, n_stars, seed=9003, truth=truth)
pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
eng = engine.Engine(pack, stars, synth.default_priors(pack_d, truth), abi.make_options())
params = synth.walker_params(truth, n_walkers, seed=42, scale=0.05)
lp = eng.logpost(params)
free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS], dtype=np.int32)
chol = np.diag([2e-4, 2e-3, 5e-4, 5e-4])
out = eng.mcmc_run_block(params, lp, np.arange(n_walkers, dtype=np.int32), free, chol, 7, 0, 40, record=False)
nw = 8192
buf = np.zeros((nw, 12), dtype=np.uint64)
eng.lib.b9_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
rc = eng.lib.b9_debug_read_stamps(buf.ctypes.data, nw); assert rc == 0
t = buf.astype(np.int64)
t = t[(t[:, 8] > 0) & (t[:, 0] > 0)]
t0 = t[:, 0].min()
names = ["entry", "loads issued", "decision", "LDS fill+barrier", "primary search", "rows+lerp", "secondary+combine", "obs/w+chi2+mix", "reduce+store"]
print(f"hot waves {len(t)}  span {t[:,8].max()-t0} ticks of 10 ns")
d = np.diff(t[:, :9], axis=1)
for k in range(8):
    print(f"  {names[k+1]:24s} mean {d[:,k].mean():9.1f}  p50 {np.median(d[:,k]):9.1f}  p95 {np.percentile(d[:,k],95):9.1f}")
dd = t[:, [1, 9, 10, 11, 2]]
for k, nm in enumerate(["partials loaded", "philox+log", "wave_sum+2 barriers", "compare"]):
    x = dd[:, k + 1] - dd[:, k]
    print(f"    decision/{nm:22s} mean {x.mean():9.1f}  p50 {np.median(x):9.1f}  p95 {np.percentile(x,95):9.1f}")
print("  wave lifetime mean", (t[:,8]-t[:,0]).mean(), " start offsets p5/p50/p95/max", np.percentile(t[:,0]-t0,[5,50,95,100]), " end offsets p50/p95/max", np.percentile(t[:,8]-t0,[50,95,100]))
