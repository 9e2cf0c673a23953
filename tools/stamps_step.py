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
# (stamps are s_memtime shader-clock counts; a wave that skipped a phase keeps an older launch's stamp there: read the medians)
print(f"hot waves {len(t)}; phase durations in shader cycles (s_memtime), last tile of each wave")
d = np.diff(t[:, :9], axis=1)
for k in range(8):
    print(f"  {names[k+1]:24s} p50 {np.median(d[:,k]):9.1f}  p95 {np.percentile(d[:,k],95):9.1f}")
dd = t[:, [1, 9, 10, 11, 2]]
for k, nm in enumerate(["partials loaded", "philox+log", "wave_sum+2 barriers", "compare"]):
    x = dd[:, k + 1] - dd[:, k]
    ok = np.abs(x) < 1e7
    if ok.any(): print(f"    decision/{nm:22s} p50 {np.median(x[ok]):9.1f}  p95 {np.percentile(x[ok],95):9.1f}")
life = t[:, 8] - t[:, 0]
print("  wave lifetime p50 %.0f p95 %.0f cycles" % tuple(np.percentile(life[np.abs(life) < 1e7], [50, 95])))
