#!/usr/bin/env python3
"""Diagnostic (build/variants/lib_mstats.so = tools/build_variant.py mstats -DB9_MARG_STATS): what the marginalised
kernel (one lane per star; a wave = 64 slot-neighbouring stars walking the node table) executes, per wave and per
star-eval, on the bench cluster; writes profiles/<tag>_marg_stats.json when given a tag.

    python tools/marg_stats.py [tag] [n_stars] [K] [Q] [walkers]
"""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["B9_HIP_LIB"] = os.path.join(ROOT, "build/variants/lib_mstats.so")
import numpy as np
from base_amd import abi, engine, synth
a = sys.argv[1:]
tag = a[0] if a and not a[0].isdigit() else None
if tag: a = a[1:]
n_stars, K, Q, W = (int(a[i]) if len(a) > i else d for i, d in enumerate((50000, 4, 4, 8)))
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, n_stars, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth),
                    abi.make_options(mode=abi.MODE_MARGINALISED, marg_iso_increm=K, marg_n_q=Q))
rows = synth.walker_params(truth, W, seed=43, scale=0.02)          # (bench.py's marginalised leg evaluates these rows)
buf = (C.c_ulonglong * 8)()
eng.lib.b9_debug_marg_stats(buf, 1)
eng.logpost(rows)
eng.lib.b9_debug_marg_stats(buf, 1)
n_evals = W * n_stars
n_wg = W * ((n_stars + 63) // 64)                  # workgroups = (64-star chunk, walker); four waves each
names = ["level-1 boxes tested (64-node chunks)", "chunk visits (a wave entering a chunk's sub-chunk)", "level-2 boxes tested (16 nodes x 1 mass ratio)",
         "units evaluated (16 terms each)", "live lane-terms (terms that enter a star's sum)"]
from base_amd import build as _build
out = {"csrc_sha256": _build.source_hash(), "n_stars": n_stars, "walkers": W, "K": K, "Q": Q, "nodes_per_star": (eng.max_eep() - 1) * K * Q,
       "rows": "synth.walker_params(truth, 8, seed=43, scale=0.02)", "counts": {}}
for k, nm in enumerate(names):
    print(f"{nm:60s} {buf[k]/n_wg:10.1f} per workgroup   {buf[k]/n_evals:9.2f} per star-eval")
    out["counts"][nm] = {"per_workgroup": buf[k] / n_wg, "per_star_eval": buf[k] / n_evals}
terms_wg = 16.0 * buf[3] / n_wg
print(f"terms evaluated per workgroup {terms_wg:.0f} (each for its 64 stars: {terms_wg:.0f} lane-terms per star-eval); live share of the lane-terms {buf[4] / (16.0 * buf[3] * 64):.3f}")
out["terms_evaluated_per_star_eval"] = terms_wg
out["live_terms_per_star_eval"] = buf[4] / n_evals
out["live_share"] = buf[4] / (16.0 * buf[3] * 64)
if tag:
    p = os.path.join(ROOT, "profiles", f"{tag}_marg_stats.json")
    json.dump(out, open(p, "w"), indent=1)
    print("wrote", p)
