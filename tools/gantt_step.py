#!/usr/bin/env python3
"""Diagnostic: Gantt chart of the fused sampler step (k_mcmc_step) from a -DB9_GANTT build: per-workgroup start / end
(s_memrealtime, 10 ns ticks) of 8 consecutive launches -> per-role timelines, launch-to-launch gaps, the tail.

    B9_HIP_LIB=build/variants/lib_gantt.so python tools/gantt_step.py [C0|C1|C2|C3|C4] [walkers]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from base_amd import abi, engine, mcmc, synth  # noqa: E402

SHAPES = {  # pack, n_filt, n_stars, wd_frac, n_y, n_pops, walkers
    "C0": ("girardi", 3, 200, 0.0, 1, 1, 1), "C1": ("dsed", 8, 10000, 0.0, 1, 1, 1), "C2": ("parsec", 8, 50000, 0.0, 1, 1, 8),
    "C3": ("parsec", 8, 20000, 0.05, 1, 1, 1), "C4": ("parsec", 8, 30000, 0.0, 3, 2, 8)}
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
pk, nf, ns, wd, ny, npops, W = SHAPES[name]
if len(sys.argv) > 2:
    W = int(sys.argv[2])
pack_d = synth.make_pack(pk, nf, n_y=ny)
truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, ns, seed=9001 + int(name[1]), truth=truth, wd_frac=wd, n_pops=npops)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth, npops), abi.make_options(n_pops=npops))
free = np.array(mcmc.DEFAULT_FREE if npops == 1 else mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA), dtype=np.int32)
start = synth.walker_params(truth, W, seed=int(os.environ.get("B9_GANTT_SEED", "7")), n_pops=npops, scale=0.02)
lp = eng.logpost(start)
chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
ids = np.arange(W, dtype=np.int32)
for _ in range(3):
    eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 200, record=False)      # warm clocks; the last 8 launches stay in the buffer
NWG = 4096
buf = np.zeros((8, NWG, 4), dtype=np.uint64)
eng.lib.b9_debug_read_gantt.argtypes = [C.c_void_p]
assert eng.lib.b9_debug_read_gantt(buf.ctypes.data) == 0
t = buf.astype(np.int64)
order = np.argsort(t[:, 0, 3])                      # by step number
steps = [t[k][t[k][:, 1] > 0] for k in order]
steps = [s for s in steps if len(s)]
print(f"{name}: {ns} stars x {nf} filters, {npops} pop(s), {W} walkers; {len(steps[0])} workgroups per launch; times in us")
ROLE = {0: "hot", 1: "heavy", 2: "derive", 3: "pad", 4: "writer"}
prev_end = None
for s in steps[1:-1]:                                # (the first and last launches of the window neighbour other kernels)
    a, b, role = s[:, 0], s[:, 1], s[:, 2] & 0xFF
    t0 = a.min()
    line = f"step {s[0, 3]}: launch span {(b.max() - t0) / 100:.2f}"
    if prev_end is not None:
        line += f"  gap after previous launch's last end {(t0 - prev_end) / 100:.2f}"
    print(line)
    prev_end = b.max()
    for r in (1, 4, 2, 0):
        m = role == r
        if not m.any():
            continue
        d = (b[m] - a[m]) / 100.0
        print(f"   {ROLE[r]:6s} n={m.sum():4d}  start p50 {np.median(a[m] - t0) / 100:6.2f} max {(a[m].max() - t0) / 100:6.2f} | "
              f"dur p50 {np.median(d):6.2f} p95 {np.percentile(d, 95):6.2f} max {d.max():6.2f} | end p50 {np.median(b[m] - t0) / 100:6.2f} "
              f"p95 {np.percentile(b[m] - t0, 95) / 100:6.2f} max {(b[m].max() - t0) / 100:6.2f}")
s = steps[len(steps) // 2]
a, b, role, xcc = s[:, 0], s[:, 1], s[:, 2] & 0xFF, (s[:, 2] >> 8) & 0xF
t0 = a.min()
last = np.argsort(b)[-6:]
hot = np.where(role == 0)[0]
dur = (b[hot] - a[hot]) / 100.0
slow = hot[dur > np.percentile(dur, 90)]
first_hot = hot.min()
print("slowest 10% of the hot workgroups (id - first hot id, xcc, us):", [(int(i - first_hot), int(xcc[i]), round((b[i] - a[i]) / 100, 1)) for i in slow][:70])
for r_, nm in ((0, "hot"), (1, "heavy"), (2, "derive")):
    m_ = role == r_
    print(f"   per XCC, {nm}: " + " ".join(f"{x}: n={int((m_ & (xcc == x)).sum())} mean {((b - a)[m_ & (xcc == x)].mean() / 100 if (m_ & (xcc == x)).any() else 0):.2f}" for x in range(8)))
if name in ("C2", "C4"):
    # placement: hot workgroups per CU (XCC, SE, SH, CU of HW_ID) against their end times
    hw = (s[:, 2] >> 16) & 0xFFFF
    cu_key = xcc * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)
    by_n = {}
    for key in np.unique(cu_key):
        m_ = cu_key == key
        n_hot, n_front = int((m_ & (role == 0)).sum()), int((m_ & (role != 0) & (role != 3)).sum())
        if n_hot:
            by_n.setdefault((n_hot, n_front), []).append((b[m_ & (role == 0)].max() - t0) / 100.0)
    print(f"   CUs in use: {len(np.unique(cu_key))}; last hot end per CU by (hot workgroups, other workgroups) on the CU:")
    for k_ in sorted(by_n):
        v_ = np.array(by_n[k_])
        print(f"      {k_[0]} hot + {k_[1]} other: {len(v_):3d} CUs, last hot end mean {v_.mean():6.2f} max {v_.max():6.2f}")
print("last finishers of a middle launch (workgroup, role, xcc, start, end):", [(int(i), ROLE[int(role[i])], int(xcc[i]), round((a[i] - t0) / 100, 2), round((b[i] - t0) / 100, 2)) for i in last])
hb = np.zeros((64, 8), dtype=np.uint64)
eng.lib.b9_debug_read_gantt_heavy.argtypes = [C.c_void_p]
if eng.lib.b9_debug_read_gantt_heavy(hb.ctypes.data) == 0:
    h = hb.astype(np.int64)
    h = h[h[:, 5] > 0]
    print("heavy role phases of the last launch, us (workgroup: entry loads + decision | LDS stage + barrier | stars | reduce):")
    for k, r in enumerate(h[:12]):
        print(f"   wg {k:2d}: loads {(r[1]-r[0])/100:5.2f} | stage+sync {(r[3]-r[1])/100:5.2f} | stars {(r[4]-r[3])/100:5.2f} | reduce {(r[5]-r[4])/100:5.2f}")
h2 = np.zeros((64, 16), dtype=np.uint64)
eng.lib.b9_debug_read_gantt_heavy2.argtypes = [C.c_void_p]
if eng.lib.b9_debug_read_gantt_heavy2(h2.ctypes.data) == 0:
    g = h2.astype(np.int64)
    hb64 = hb.astype(np.int64)
    print("inside one heavy star (lane HS2_LANE of wave 0), us since the role's stars phase began (heavy stamp 3): "
          "entry 1 | MS search 11-12 | WD branch entered 3 | prec 2 | exp/log 4 | cooling: axes 13, age 14, done 6 | desc 7 | chi2 done 8")
    for k in range(12):
        if g[k, 7] <= 0 and g[k, 12] <= 0:
            continue
        t0 = hb64[k, 3]
        print("   wg %2d: " % k + " ".join("%d:%5.2f" % (j, (g[k, j] - t0) / 100) if g[k, j] else "%d:  -  " % j for j in (1, 11, 12, 3, 2, 4, 13, 14, 6, 7, 8))
              + "   | stars phase ends %5.2f" % ((hb64[k, 4] - t0) / 100))
wk = np.zeros(8, dtype=np.uint64)
if hasattr(eng.lib, "b9_debug_read_gantt_walk"):
    eng.lib.b9_debug_read_gantt_walk.argtypes = [C.c_void_p]
    if eng.lib.b9_debug_read_gantt_walk(wk.ctypes.data) == 0 and wk[4] > 0:
        t = wk.astype(np.int64)
        print("tree walk of workgroup 0 (the writer), us since kernel entry: loads issued from %.2f to %.2f | landed %.2f | walk done %.2f"
              % ((t[1] - t[0]) / 100, (t[2] - t[0]) / 100, (t[3] - t[0]) / 100, (t[4] - t[0]) / 100))
