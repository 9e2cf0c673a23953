#!/usr/bin/env python3
"""Throughput of b9_sample_mass (the sampleMass counterpart) next to the plain marginalised log-posterior,
50k stars x 8 filters, 4 sub-steps x 4 mass ratios (6384 nodes per star and row)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
opt = abi.make_options(mode=abi.MODE_MARGINALISED, marg_iso_increm=4, marg_n_q=4)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), opt)
rows = synth.walker_params(truth, 32, seed=42, scale=0.05)
eng.logpost(rows[:8]); eng.sample_mass(rows[:8])
t0 = time.perf_counter(); eng.logpost(rows); t_lp = time.perf_counter() - t0
t0 = time.perf_counter(); m, q, mem, pop = eng.sample_mass(rows, seed=3); t_sm = time.perf_counter() - t0
print(f"marginalised logpost: {32*50000/t_lp:.3e} star-evals/s;  sample_mass: {32*50000/t_sm:.3e} star draws/s ({t_sm/t_lp:.2f}x the time)")
print("median |sampled mass - catalogue| / catalogue:", np.median(np.abs(np.median(m, axis=0) - cl['mass1']) / cl['mass1']),
      " mean membership of members:", mem.mean())
