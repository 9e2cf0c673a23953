"""Diagnostic (B9_HIP_LIB=build/variants/lib_gantt.so): are the last hot workgroups of the bench shape's launch the same ones every
launch, and what do they have in common?  Prints the mean end time by tile group and by walker over six consecutive launches."""
import os, sys, ctypes as C
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
for _ in range(3): eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 200, record=False)
buf = np.zeros((8, 4096, 4), dtype=np.uint64)
eng.lib.b9_debug_read_gantt.argtypes = [C.c_void_p]
assert eng.lib.b9_debug_read_gantt(buf.ctypes.data) == 0
t = buf.astype(np.int64)
ends = []
for k in range(8):
    s = t[k][t[k][:, 1] > 0]
    role = s[:, 2] & 0xFF
    hot = np.where(role == 0)[0]
    t0 = s[:, 0].min()
    ends.append((s[hot, 1] - t0) / 100.0)
n = min(len(e) for e in ends)
E = np.array([e[:n] for e in ends[1:-1]])          # launches x hot workgroups
L = np.arange(n); xcd = L & 7; sidx = L >> 3; w = sidx % 8; group = (sidx // 8) * 8 + xcd
m = E.mean(axis=0)
print("hot workgroups", n, "mean end", m.mean().round(2), "max of per-WG means", m.max().round(2), "mean of per-launch max", E.max(axis=1).mean().round(2))
print("correlation of a workgroup's end time between consecutive launches:", np.corrcoef(E[0], E[1])[0, 1].round(3), np.corrcoef(E[2], E[3])[0, 1].round(3))
bygroup = np.array([m[group == g].mean() for g in range(group.max() + 1)])
print("mean end by tile group (66 groups):", np.round(bygroup, 1).tolist())
byw = [m[w == k].mean().round(2) for k in range(8)]
print("mean end by walker:", byw)
