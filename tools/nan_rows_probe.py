import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle
from base_amd import abi, engine, synth, mcmc
from conftest import build_problem
for npops, mode in ((1, abi.MODE_GIVEN_MASS), (2, abi.MODE_GIVEN_MASS), (1, abi.MODE_MARGINALISED)):
    pack_d, cl, pack, stars, priors, _ = build_problem("dsed", 5, n_stars=300, wd_frac=0.1, n_y=3 if npops == 2 else 1, n_pops=npops, seed=4)
    opt = abi.make_options(mode=mode, n_pops=npops, marg_iso_increm=2, marg_n_q=2)
    eng = engine.Engine(pack, stars, priors, opt); orc = oracle.Oracle(pack, stars, priors, opt)
    rows = []
    for k in range(abi.B9_NPARAM):
        for bad in (np.nan, np.inf, -np.inf, 1e300, -1e300):
            r = cl["truth"].copy(); r[k] = bad; rows.append(r)
    rows = np.array(rows)
    got = np.concatenate([eng.logpost(rows[i:i+20]) for i in range(0, len(rows), 20)])
    want = orc.logpost(rows)
    same = np.array_equal(np.isfinite(got), np.isfinite(want)) and np.array_equal(np.isnan(got), np.isnan(want))
    fin = np.isfinite(want)
    err = np.max(np.abs(got[fin] - want[fin]) / np.maximum(1, np.abs(want[fin]))) if fin.any() else 0
    print(f"pops {npops} mode {mode}: rows {len(rows)} finite {fin.sum()} nan(gpu) {np.isnan(got).sum()} nan(oracle) {np.isnan(want).sum()} same-support {same} max err {err:.2e}")
    if not same:
        bad = np.where((np.isfinite(got) != np.isfinite(want)) | (np.isnan(got) != np.isnan(want)))[0]
        for b in bad[:10]: print("   row", b, "param", b // 5, "value", rows[b, b // 5], "gpu", got[b], "oracle", want[b])
    if mode == abi.MODE_GIVEN_MASS:   # the sampler must survive a NaN/inf-producing proposal factor too
        free = np.array(mcmc.DEFAULT_FREE); chol = np.diag([1e300, 1e-3, np.inf, 1e-3])
        start = synth.walker_params(cl["truth"], 4, seed=1, n_pops=npops)
        out = eng.mcmc_run_block(start, eng.logpost(start), np.arange(4), free, chol, 1, 0, 20)
        print("   sampler with inf/1e300 steps: accepted", out[4], "state finite", bool(np.all(np.isfinite(out[0]))), "lp finite", bool(np.all(np.isfinite(out[1]))))
