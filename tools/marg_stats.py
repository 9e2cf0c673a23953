#!/usr/bin/env python3
"""Diagnostic (build/variants/lib_mstats.so, -DB9_MARG_STATS): how many 64-node chunks / mass-ratio iterations /
filter evaluations the marginalised kernel executes per star."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["B9_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build/variants/lib_mstats.so")
import numpy as np
from base_amd import abi, engine, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth),
                    abi.make_options(mode=abi.MODE_MARGINALISED, marg_iso_increm=4, marg_n_q=4))
rows = synth.walker_params(truth, 8, seed=42, scale=0.05)
buf = (C.c_ulonglong * 8)()
eng.lib.b9_debug_marg_stats(buf, 1)
eng.logpost(rows)
eng.lib.b9_debug_marg_stats(buf, 1)
n = 8 * 50000
names = ["chunks visited", "chunks past the chunk bound", "chunks with a live node", "chunks entering the mass-ratio loop",
         "mass-ratio iterations", "filter evaluations in them", "lanes wanting the mass-ratio loop (sum)"]
for k, nm in enumerate(names):
    print(f"{nm:45s} {buf[k]/n:10.2f} per star")
