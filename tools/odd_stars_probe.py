#!/usr/bin/env python3
"""Probe: odd star records -- the ones b9_load_stars must reject (NaN / inf masses and observations, mass ratios outside [0, 1],
priors outside (0, 1], ...) are each rejected with their reason; the odd ones it accepts (1e300 observations, unused filters, tiny
sigmas, huge / tiny masses, certain members) go through the HIP path against the oracle, per star, both modes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from base_amd import abi, engine, synth
from conftest import build_problem
rc = 0
# (1) records the loader must REJECT: each one alone, with its reason
pack_d, cl0, pack, stars, priors, _ = build_problem("parsec", 4, n_stars=64, wd_frac=0.1, seed=4)
def variant(edit):
    c = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in cl0.items()}
    edit(c)
    return abi.make_stars(c)
def set_(key, i, v):
    def f(c): c[key].reshape(-1)[i] = v
    return f
rejected = [("NaN mass", set_("mass1", 9, np.nan)), ("inf mass", set_("mass1", 13, np.inf)), ("mass ratio 1.5", set_("mass_ratio", 16, 1.5)),
            ("mass ratio -0.3", set_("mass_ratio", 17, -0.3)), ("NaN mass ratio", set_("mass_ratio", 18, np.nan)),
            ("NaN observation", set_("obs", 12, np.nan)), ("inf observation", set_("obs", 12, np.inf)), ("NaN sigma", set_("sigma", 7, np.nan)),
            ("sigma 1e-200", set_("sigma", 7, 1e-200)), ("prior 0", set_("clust_prior", 20, 0.0)), ("prior 1.5", set_("clust_prior", 20, 1.5))]
for name, edit in rejected:
    try:
        engine.Engine(pack, variant(edit), priors, abi.make_options())
        print(f"NOT rejected: {name}"); rc |= 1
    except engine.B9Error as e:
        print(f"rejected as it should be: {name:18s} -> {str(e)[:90]}")
try:    # given-mass mode needs positive masses (checked at the first evaluation); the marginalised mode takes them as hints
    e_ = engine.Engine(pack, variant(set_("mass1", 10, 0.0)), priors, abi.make_options()); e_.logpost(synth.walker_params(cl0["truth"], 1, seed=2))
    print("NOT rejected: mass 0 in given-mass mode"); rc |= 1
except engine.B9Error as e:
    print(f"rejected as it should be: mass 0, given-mass -> {str(e)[:90]}")
# (2) odd records the loader accepts: the HIP path against the oracle, per star
for mode in (abi.MODE_GIVEN_MASS, abi.MODE_MARGINALISED):
    cl = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in cl0.items()}
    obs, sig = cl["obs"].reshape(64, 4), cl["sigma"].reshape(64, 4)
    obs[3, :] = 1e300
    sig[4, 0] = 0.0; sig[5, :] = -1.0; sig[6, 2] = 1e-140
    m, q, pm = cl["mass1"], cl["mass_ratio"], cl["clust_prior"]
    m[12] = 1e9; m[14] = 1e-9
    if mode == abi.MODE_MARGINALISED: m[10] = 0.0; m[11] = -1.0
    q[15] = 1.0; q[19] = 1e-12
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
