#!/usr/bin/env python3
"""Diagnostic: Gantt chart of the marginalised mode's fused sampler step (k_marg_step) from a -DB9_GANTT build: per-workgroup
start / end (s_memrealtime, 10 ns ticks) of consecutive launches -> per-role timelines, and the table builders' phases.

    B9_HIP_LIB=build/variants/lib_gantt.so python tools/gantt_marg.py [C0|C1|C2|C3|C4] [K Q]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from base_amd import abi, engine, mcmc, synth  # noqa: E402

SHAPES = {"C0": ("girardi", 3, 200, 0.0, 1, 1, 1), "C1": ("dsed", 8, 10000, 0.0, 1, 1, 1), "C2": ("parsec", 8, 50000, 0.0, 1, 1, 8),
          "C3": ("parsec", 8, 20000, 0.05, 1, 1, 1), "C4": ("parsec", 8, 30000, 0.0, 3, 2, 8)}
name = sys.argv[1] if len(sys.argv) > 1 else "C1"
K, Q = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 4)
pk, nf, ns, wd, ny, npops, W = SHAPES[name]
pack_d = synth.make_pack(pk, nf, n_y=ny)
truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, ns, seed=9001 + int(name[1]), truth=truth, wd_frac=wd, n_pops=npops)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth, npops), abi.make_options(abi.MODE_MARGINALISED, npops, K, Q))
free = np.array(mcmc.DEFAULT_FREE if npops == 1 else mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA), dtype=np.int32)
start = synth.walker_params(truth, W, seed=7, n_pops=npops, scale=0.02)
lp = eng.logpost(start)
chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
ids = np.arange(W, dtype=np.int32)
for _ in range(3):
    eng.mcmc_run_block(start, lp, ids, free, chol, 7, 0, 200, record=False)
NWG = 8192
buf = np.zeros((8, NWG, 4), dtype=np.uint64)
eng.lib.b9_debug_read_gantt.argtypes = [C.c_void_p]
assert eng.lib.b9_debug_read_gantt(buf.ctypes.data) == 0
t = buf.astype(np.int64)
order = np.argsort(t[:, 0, 3])
steps = [t[k][t[k][:, 1] > 0] for k in order]
steps = [s for s in steps if len(s)]
print(f"{name} marginalised {K} x {Q}: {ns} stars x {nf} filters, {npops} pop(s), {W} walkers; {len(steps[0])} workgroups recorded per launch (of the first {NWG}); us")
ROLE = {0: "stars", 1: "wd-stars", 2: "table", 3: "pad", 4: "writer", 5: "wd-table"}
prev_end = None
for s in steps[1:-1]:
    a, b, role = s[:, 0], s[:, 1], s[:, 2] & 0xFF
    t0 = a.min()
    line = f"step {s[0, 3]}: launch span {(b.max() - t0) / 100:.2f}"
    if prev_end is not None:
        line += f"  gap after the previous launch's last end {(t0 - prev_end) / 100:.2f}"
    print(line)
    prev_end = b.max()
    # workgroups in flight over the launch (tenths of its span)
    edges = np.linspace(t0, b.max(), 11)
    busy = [float(np.clip(np.minimum(b, edges[k + 1]) - np.maximum(a, edges[k]), 0, None).sum() / (edges[k + 1] - edges[k])) for k in range(10)]
    print("   workgroups in flight, by tenth of the span: " + " ".join(f"{x:5.0f}" for x in busy) + f"   (mean {np.sum(b - a) / (b.max() - t0):.0f})")
    for r in (4, 2, 5, 0, 1):
        m = role == r
        if not m.any():
            continue
        d = (b[m] - a[m]) / 100.0
        print(f"   {ROLE[r]:8s} n={m.sum():4d}  start p50 {np.median(a[m] - t0) / 100:6.2f} max {(a[m].max() - t0) / 100:6.2f} | "
              f"dur p50 {np.median(d):6.2f} p95 {np.percentile(d, 95):6.2f} max {d.max():6.2f} | end p50 {np.median(b[m] - t0) / 100:6.2f} "
              f"p95 {np.percentile(b[m] - t0, 95) / 100:6.2f} max {(b[m].max() - t0) / 100:6.2f}")
hb = np.zeros((64, 8), dtype=np.uint64)
eng.lib.b9_debug_read_gantt_heavy.argtypes = [C.c_void_p]
if eng.lib.b9_debug_read_gantt_heavy(hb.ctypes.data) == 0:
    h = hb.astype(np.int64)
    print("table builders' phases of the last launch, us (workgroup: candidate row | header | tiles | rows | nb + box1):")
    for k in range(64):
        r = h[k]
        if r[5] <= 0 or r[0] <= 0:
            continue
        print(f"   wg {k:2d}: row {(r[1]-r[0])/100:5.2f} | header {(r[2]-r[1])/100:5.2f} | tiles {(r[3]-r[2])/100:5.2f} | rows {(r[4]-r[3])/100:5.2f} | nb+box1 {(r[5]-r[4])/100:5.2f} | total {(r[5]-r[0])/100:5.2f}")
sb = np.zeros((64, 4, 8), dtype=np.uint64)
if hasattr(eng.lib, "b9_debug_read_gantt_star"):
    eng.lib.b9_debug_read_gantt_star.argtypes = [C.c_void_p]
    if eng.lib.b9_debug_read_gantt_star(sb.ctypes.data) == 0:
        h = sb.astype(np.int64)
        print("star workgroups, dispatch positions 0-63 (the launch's most expensive pieces), last launch, us: entry -> decision | headers + level 1 + barrier | "
              "walk of waves 0-3 | wave 0's wait for the others | merge")
        for k in range(64):
            r = h[k][0]
            if r[0] <= 0 or r[4] <= 0 or r[3] < r[2]:
                continue
            walks = " ".join(f"{(h[k][v][3]-h[k][v][2])/100:5.2f}" for v in range(4))
            print(f"   wg {k:2d}: decision {(r[1]-r[0])/100:5.2f} | level 1 {(r[2]-r[1])/100:5.2f} | walk {walks} | barrier {(r[4]-r[3])/100:5.2f} | merge {(r[5]-r[4])/100:5.2f} | total {(r[5]-r[0])/100:5.2f}")
