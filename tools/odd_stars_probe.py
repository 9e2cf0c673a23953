#!/usr/bin/env python3
"""Probe: odd star records (NaN / inf observations, zero / negative sigmas, NaN / zero / huge masses, mass ratios
outside [0,1), certain members) -- the HIP path against the oracle, per star, both modes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from base_amd import abi, engine, synth
from conftest import build_problem
rc = 0
for mode in (abi.MODE_GIVEN_MASS, abi.MODE_MARGINALISED):
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", 4, n_stars=64, wd_frac=0.1, seed=4)
    cl = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in cl.items()}
    obs, sig = cl["obs"].reshape(64, 4), cl["sigma"].reshape(64, 4)
    obs[3, :] = 1e300                                     # (NaN / inf observations and sigmas are rejected by b9_load_stars)
    sig[4, 0] = 0.0; sig[5, :] = -1.0; sig[6, 2] = 1e-140
    m, q, pm = cl["mass1"], cl["mass_ratio"], cl["clust_prior"]
    m[9] = np.nan; m[10] = 0.0; m[11] = -1.0; m[12] = 1e9; m[13] = np.inf; m[14] = 1e-9
    q[15] = 1.0; q[16] = 1.5; q[17] = -0.3; q[18] = np.nan; q[19] = 1e-12
    pm[20] = 1.0; pm[21] = 1e-300
    cl["obs"], cl["sigma"] = obs.reshape(-1), sig.reshape(-1)
    st = abi.make_stars(cl)
    opt = abi.make_options(mode=mode, marg_iso_increm=2, marg_n_q=3)
    eng, orc = engine.Engine(pack, st, priors, opt), oracle.Oracle(pack, st, priors, opt)
    rows = synth.walker_params(cl["truth"], 3, seed=2)
    g, gp = eng.logpost(rows, perstar=True); w, wp = orc.logpost(rows, perstar=True)
    bad = (np.isnan(gp) != np.isnan(wp)) | (np.isfinite(gp) != np.isfinite(wp))
    fin = np.isfinite(wp) & ~bad
    err = np.max(np.abs(gp[fin] - wp[fin]) / np.maximum(1, np.abs(wp[fin])))
    print(f"mode {mode}: stars with a different verdict {sorted(set(np.where(bad)[1]))}  max err elsewhere {err:.2e}  totals gpu {g} oracle {w}")
    for s in sorted(set(np.where(bad)[1])): print("    star", s, "gpu", gp[:, s], "oracle", wp[:, s])
    rc |= int(bad.any() or err > 1e-9)
sys.exit(rc)
