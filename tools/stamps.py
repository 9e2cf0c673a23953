#!/usr/bin/env python3
"""Diagnostic: per-phase wave timeline of k_star_like from the -DB9_STAMPS build."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["B9_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build/variants/lib_stamps.so")
import numpy as np, torch
from base_amd import abi, engine, synth
n_stars, n_walkers = 50000, 8
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, n_stars, seed=9003, truth=truth)
pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
eng = engine.Engine(pack, stars, synth.default_priors(pack_d, truth), abi.make_options())
params = synth.walker_params(truth, n_walkers, seed=42, scale=0.05)
for _ in range(5): eng.logpost(params)
nw = 6272
buf = np.zeros((nw, 12), dtype=np.uint64)
eng.lib.b9_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
rc = eng.lib.b9_debug_read_stamps(buf.ctypes.data, nw); assert rc == 0
t = buf.astype(np.int64)
ok = t[:, 8] > 0
t = t[ok]
t0 = t[:, 0].min()
names = ["entry", "hdr/params", "mass->LDS+barrier", "star scalars", "primary search+rows", "secondary+combine", "obs/w+chi2", "mixture", "reduce+store"]
print(f"waves {len(t)}  kernel span raw ticks {t[:,8].max()-t0}")
d = np.diff(t[:, :9], axis=1)
for k in range(8):
    print(f"  {names[k+1]:24s} mean {d[:,k].mean():9.1f}  p50 {np.median(d[:,k]):9.1f}  p95 {np.percentile(d[:,k],95):9.1f}")
print("  wave lifetime mean", (t[:,8]-t[:,0]).mean(), " start offsets p50/p95/max", np.percentile(t[:,0]-t0,[50,95,100]))
