#!/usr/bin/env python3
"""Diagnostic (lib_mlife.so): start / end of every k_star_marg workgroup of one call on the bench shape."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["B9_HIP_LIB"] = os.path.join(ROOT, "build/variants/lib_mlife.so")
import numpy as np
from base_amd import abi, engine, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth),
                    abi.make_options(mode=abi.MODE_MARGINALISED, marg_iso_increm=4, marg_n_q=4))
rows = synth.walker_params(truth, 8, seed=42, scale=0.05)
eng.logpost(rows); eng.logpost(rows)
buf = (C.c_ulonglong * (16384 * 4))()
eng.lib.b9_debug_marg_life(buf)
a = np.array(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en, units = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2] / 2     # us; (two calls accumulated the unit counts)
print(f"{len(a)} workgroups; launch span {en.max():.1f} us; life mean {np.mean(en - st):.1f} median {np.median(en - st):.1f} p99 {np.percentile(en - st, 99):.1f} max {np.max(en - st):.1f} us")
print("start times: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(st, [50, 90, 99, 100])))
print("units of wave 0: mean %.1f p99 %.1f max %.0f" % (units.mean(), np.percentile(units, 99), units.max()))
order = np.argsort(-(en - st))[:12]
for i in order: print(f"  wg {i:5d} xcd {i % 8} start {st[i]:7.1f} end {en[i]:7.1f} life {en[i]-st[i]:6.1f} units(w0) {units[i]:.0f}")
for lo in range(0, int(en.max()) + 1, 20):
    print(f"  t={lo:4d}us resident {np.sum((st <= lo) & (en > lo)):5d}")
