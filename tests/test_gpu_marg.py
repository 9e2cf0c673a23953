"""GPU parity of the marginalised mode (k_star_marg: one lane per star, four waves per 64 stars) against the oracle's
brute-force definition of the same integral (oracle/b9_oracle.c::star_marg_loglike).
[RECALL] BASE-9's marg.cpp restricts the secondary-mass range adaptively; that cannot be restated
without the source, so both sides integrate the full (mass, mass-ratio) grid -- parity unpinned."""
import numpy as np
import pytest

import oracle
from base_amd import abi, synth
from conftest import build_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_filt,n_y,n_pops,wd_frac,K,Q", [(8, 1, 1, 0.0, 3, 3), (3, 1, 1, 0.15, 2, 4), (8, 3, 2, 0.1, 2, 2), (5, 1, 1, 0.0, 1, 1)])
def test_marginalised_matches_oracle(n_filt, n_y, n_pops, wd_frac, K, Q):
    from base_amd import engine
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", n_filt, n_stars=90, wd_frac=wd_frac, n_y=n_y,
                                                       n_pops=n_pops, n_feh=3, n_age=5, n_eep=40)
    options = abi.make_options(abi.MODE_MARGINALISED, n_pops, K, Q)
    eng = engine.Engine(pack, stars, priors, options)
    orc = oracle.Oracle(pack, stars, priors, options)
    params = synth.walker_params(cl["truth"], 3, n_pops=n_pops)
    params[2, abi.P_LOGAGE] = pack_d["log_age"][-1] + 1.0           # outside the grid
    lp_g, ps_g = eng.logpost(params, perstar=True)
    lp_o, ps_o = orc.logpost(params, perstar=True)
    assert lp_g[2] == -np.inf and np.all(ps_g[2] == -np.inf)
    fin = np.isfinite(ps_o)
    assert np.array_equal(np.isfinite(ps_g), fin)
    err = np.abs(ps_g[fin] - ps_o[fin]) / np.maximum(1.0, np.abs(ps_o[fin]))
    assert err.max() <= 1e-9, err.max()
    assert np.all(np.abs(lp_g[:2] - lp_o[:2]) <= 1e-9 * np.maximum(1.0, np.abs(lp_o[:2])))
    # the marginal likelihood of a star is never below its best single node's contribution, and the
    # mode switch really changes the answer
    given = engine.Engine(pack, stars, priors, abi.make_options(abi.MODE_GIVEN_MASS, n_pops)).logpost(params[:1])
    assert abs(given[0] - lp_g[0]) > 1e-3


def test_marginalised_mcmc_block_runs():
    from base_amd import engine, mcmc
    pack_d, cl, pack, stars, priors, _ = build_problem("dsed", 8, n_stars=64, n_feh=3, n_age=5, n_eep=40)
    eng = engine.Engine(pack, stars, priors, abi.make_options(abi.MODE_MARGINALISED, 1, 2, 2))
    start = synth.walker_params(cl["truth"], 4, scale=0.1)
    s = mcmc.WalkerSampler(start, mcmc.DeviceBlockRunner(eng), block=20, seed=3)
    s.initialise(eng.logpost)
    s.run(60)
    assert np.all(np.isfinite(s.all_logpost)) and s.accepted > 0


@pytest.mark.parametrize("n_pops", [1, 2])
def test_marginalised_device_block_matches_host_twin(n_pops):
    """The two-launch sampler step (what marginalised mode runs) against the host twin driving the same
    marginalised log-posterior: same chain."""
    from base_amd import engine, mcmc
    pack_d, cl, pack, stars, priors, _ = build_problem("dsed", 5, n_stars=120, wd_frac=0.05, n_y=3 if n_pops == 2 else 1,
                                                       n_pops=n_pops, n_feh=3, n_age=5, n_eep=40, seed=6)
    eng = engine.Engine(pack, stars, priors, abi.make_options(abi.MODE_MARGINALISED, n_pops, 2, 3))
    free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS] + ([abi.P_Y, abi.P_Y2, abi.P_LAMBDA] if n_pops == 2 else []))
    chol = np.diag([5e-4, 3e-3, 1e-3, 1e-3] + ([4e-4, 4e-4, 3e-3] if n_pops == 2 else []))
    start = synth.walker_params(cl["truth"], 5, seed=2, scale=0.1, n_pops=n_pops)
    lp0 = eng.logpost(start)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, np.arange(5), free, chol, 21, 500, 15)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, np.arange(5), free, chol, 21, 500, 15)
    assert dev[4] == host[4] and 0 < dev[4] < 75
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)
    np.testing.assert_allclose(dev[1], oracle.Oracle(pack, stars, priors, abi.make_options(abi.MODE_MARGINALISED, n_pops, 2, 3)).logpost(dev[0]), rtol=1e-9)


@pytest.mark.parametrize("name", ["C1", "C2", "C3", "C4"])
def test_marginalised_matches_oracle_full_size(name):
    """The marginalised mode at the FULL size of the BASELINE.json configurations (K = Q = 4: 6384 nodes per star and
    population): one parameter row, every per-star value and the total against the oracle's brute-force integral at
    1e-9 relative.  C3 holds WD-stage stars (k_star_marg_wd beside k_star_marg), C4 two populations.  The oracle runs its
    OpenMP build over the stars (same per-star values as the sequential checker; a few seconds per row)."""
    from base_amd import engine
    cfg = synth.make_baseline_config(name)
    n_pops = cfg["n_pops"]
    opt = abi.make_options(abi.MODE_MARGINALISED, n_pops, 4, 4)
    eng = engine.Engine(cfg["pack"], cfg["stars"], cfg["priors"], opt)
    row = synth.walker_params(cfg["truth"], 3, seed=5, n_pops=n_pops, scale=0.05)[2:3]
    lp_g, ps_g = eng.logpost(row, perstar=True)
    lp_o, ps_o = oracle.Oracle(cfg["pack"], cfg["stars"], cfg["priors"], opt, native=True).logpost(row, perstar=True)
    stage = np.asarray(cfg["cluster"]["stage"])
    if name == "C3":
        assert (stage == abi.STAGE_WD).sum() > 500
    if name == "C4":
        assert n_pops == 2
    assert np.all(np.isfinite(ps_o)) and np.array_equal(np.isfinite(ps_g), np.isfinite(ps_o))
    err = np.abs(ps_g - ps_o) / np.maximum(1.0, np.abs(ps_o))
    assert err.max() <= 1e-9, (name, err.max(), int(err.argmax()))
    assert abs(lp_g[0] - lp_o[0]) <= 1e-9 * max(1.0, abs(lp_o[0])), (name, lp_g[0], lp_o[0])
    eng.close()


@pytest.mark.parametrize("n_pops,wd_frac,K,Q", [(1, 0.0, 4, 4), (2, 0.05, 2, 3), (1, 0.0, 1, 8)])
def test_pruning_is_rigorous(n_pops, wd_frac, K, Q):
    """The marginalised kernel with its pruning (field floor, two levels of boxes, running maxima shared between waves)
    against THE SAME kernel evaluating every node of every star (b9_tuning.marg_no_pruning): equal to 1e-12 -- what the
    pruning drops is below e^-40 of what it keeps.  Includes stars the floor prunes entirely (field stars), membership
    priors of exactly 1 (no floor: the seed pass) and of 1e-200."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", 8, n_stars=6000, wd_frac=wd_frac, n_y=3 if n_pops == 2 else 1,
                                                       n_pops=n_pops, small=False, seed=77)
    cl["clust_prior"][:200] = 1.0
    cl["clust_prior"][200:260] = 1e-200
    # photometry 50 x sharper than the grid resolves (scaled observations of ~1e5: the packed-fp32 box test of the one-population
    # instances runs on its rounding slack there), with and without a field floor
    sg = np.asarray(cl["sigma"]).reshape(len(cl["clust_prior"]), -1)
    sg[260:420] = np.where(sg[260:420] > 0, sg[260:420] * 0.02, sg[260:420])
    cl["sigma"] = sg.reshape(np.asarray(cl["sigma"]).shape)
    cl["clust_prior"][260:330] = 1.0
    stars = abi.make_stars(cl)
    opt = abi.make_options(abi.MODE_MARGINALISED, n_pops, K, Q)
    eng = engine.Engine(pack, stars, priors, opt)
    rows = synth.walker_params(cl["truth"], 4, seed=9, n_pops=n_pops, scale=0.3)
    lp_a, ps_a = eng.logpost(rows, perstar=True)
    eng.set_tuning(marg_no_pruning=1)
    lp_b, ps_b = eng.logpost(rows, perstar=True)
    eng.set_tuning()
    assert np.array_equal(np.isfinite(ps_a), np.isfinite(ps_b))
    fin = np.isfinite(ps_b)
    assert fin.mean() > 0.9
    err = np.abs(ps_a[fin] - ps_b[fin]) / np.maximum(1.0, np.abs(ps_b[fin]))
    assert err.max() <= 1e-12, err.max()
    np.testing.assert_allclose(lp_a, lp_b, rtol=1e-12)
    eng.close()


@pytest.mark.parametrize("name,walkers", [("C1", 1), ("C2", 8)])
def test_marginalised_mode_is_bit_reproducible(name, walkers):
    """Which terms enter a star's sum is a function of the data only (the pruning reference is the barrier-merged seed maxima
    plus the wave's OWN running maximum), so repeated evaluations return the same bits for every star -- on the split
    instance (C1: 157 star chunks, a chunk's window over 8 workgroups + k_marg_merge) and the unsplit one (C2) -- and two
    runs of the fused sampler step (k_marg_step) give the same chain.  tools/soak_determinism.py --marg is the long version."""
    from base_amd import engine, mcmc
    cfg = synth.make_baseline_config(name)
    eng = engine.Engine(cfg["pack"], cfg["stars"], cfg["priors"], abi.make_options(abi.MODE_MARGINALISED, 1, 4, 4))
    rows = synth.walker_params(cfg["truth"], walkers, seed=11, scale=0.03)
    lp0, ps0 = eng.logpost(rows, perstar=True)
    assert np.all(np.isfinite(ps0))
    for _ in range(20):
        lp, ps = eng.logpost(rows, perstar=True)
        assert np.array_equal(lp, lp0) and np.array_equal(ps, ps0)
    free = np.array(mcmc.DEFAULT_FREE)
    chol = np.diag([2e-5, 1e-4, 4e-5, 4e-5]) * 3.0
    runs = [eng.mcmc_run_block(rows, lp0, np.arange(walkers), free, chol, 5, 0, 60) for _ in range(2)]
    for a, b in zip(runs[0][:4], runs[1][:4]):
        assert np.array_equal(a, b)
    assert runs[0][4] == runs[1][4] and 0 < runs[0][4] < 60 * walkers
    eng.close()


@pytest.mark.parametrize("n_pops,wd_frac,n_stars,walkers", [(1, 0.0, 9000, 3), (1, 0.08, 2500, 2), (2, 0.05, 3000, 2), (1, 0.0, 40000, 2)])
def test_fused_marginalised_step_equals_the_two_launch_step(n_pops, wd_frac, n_stars, walkers):
    """k_marg_step (one launch per step: decision + stars + both candidate node tables of the next step, built by workgroups
    that derive their isochrone tiles themselves) against the two-launch step it replaced (k_derive_iso, k_marg_table,
    k_star_marg [, k_marg_merge, k_marg_wd_table, k_star_marg_wd]; b9_tuning.two_launch_steps): the same chain -- same
    proposals (bit for bit), the same decisions, log-posteriors equal to the rounding of the two decision sums' orders.
    Covers split catalogues (pieces + merge), WD-stage stars, two populations with a helium axis (eight corner isochrones per
    table value), an unsplit catalogue, continued blocks, and proposals that leave the grid (a step scale that reaches the
    grid's edge in log age)."""
    from base_amd import engine, mcmc
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", 8, n_stars=n_stars, wd_frac=wd_frac, n_y=3 if n_pops == 2 else 1,
                                                       n_pops=n_pops, small=False, seed=21)
    eng = engine.Engine(pack, stars, priors, abi.make_options(abi.MODE_MARGINALISED, n_pops, 3, 3))
    free = np.array(list(mcmc.DEFAULT_FREE) + ([abi.P_Y, abi.P_Y2, abi.P_LAMBDA] if n_pops == 2 else []))
    chol = np.diag([3e-4, 2e-3, 8e-4, 8e-4] + ([3e-4, 3e-4, 2e-3] if n_pops == 2 else []))
    chol[0, 0] = 0.4 if n_stars == 2500 else chol[0, 0]                 # (this case: most log-age proposals fall off the grid)
    start = synth.walker_params(cl["truth"], walkers, seed=3, scale=0.05, n_pops=n_pops)
    lp0 = eng.logpost(start)
    ids = np.arange(walkers)

    def chain():
        a = eng.mcmc_run_block(start, lp0, ids, free, chol, 17, 0, 40)
        b = eng.mcmc_run_block(a[0], a[1], ids, free, chol, 17, 40, 25)
        return a, b
    fused = chain()
    eng.set_tuning(two_launch_steps=1)
    two = chain()
    eng.set_tuning()
    for f, t in zip(fused, two):
        assert f[4] == t[4]
        np.testing.assert_array_equal(f[2], t[2])                        # the chain: identical positions
        np.testing.assert_allclose(f[3], t[3], rtol=1e-12)              # its log-posteriors
        np.testing.assert_array_equal(f[0], t[0])
    assert 0 < fused[0][4] < 40 * walkers
    eng.close()


def test_catalogue_plan_changes_rounding_only_and_survives_an_off_grid_reference():
    """The marginalised catalogue plan (measured dispatch order, cost-proportional pieces of a small catalogue) decides how a
    star's sum is grouped, nothing else: per-star values under different piece sizes (b9_tuning.marg_piece_units) agree to
    1e-12 and each setting is bit-reproducible; priors whose means lie OUTSIDE the grid (the plan's reference row is clamped
    into it) and priors without a usable reference (NaN means) still give the oracle's values."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", 8, n_stars=5000, small=False, seed=31)
    opt = abi.make_options(abi.MODE_MARGINALISED, 1, 3, 4)
    eng = engine.Engine(pack, stars, priors, opt)
    rows = synth.walker_params(cl["truth"], 2, seed=5, scale=0.05)
    base_lp, base_ps = eng.logpost(rows, perstar=True)
    for units in (2, 9):
        eng.set_tuning(marg_piece_units=units)
        lp, ps = eng.logpost(rows, perstar=True)
        lp2, ps2 = eng.logpost(rows, perstar=True)
        assert np.array_equal(ps, ps2) and np.array_equal(lp, lp2)
        np.testing.assert_allclose(ps, base_ps, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(lp, base_lp, rtol=1e-12)
    eng.set_tuning()
    want = oracle.Oracle(pack, stars, priors, opt, native=True).logpost(rows[:1], perstar=True)[1]
    for bad in ("outside", "nan"):
        pr = abi.b9_priors()
        for k in range(abi.B9_NPARAM):
            pr.mean[k], pr.var[k] = priors.mean[k], priors.var[k]
        pr.log_age_min, pr.log_age_max = priors.log_age_min, priors.log_age_max
        pr.mean[abi.P_LOGAGE] = 99.0 if bad == "outside" else float("nan")
        pr.mean[abi.P_MOD] = priors.mean[abi.P_MOD] + (25.0 if bad == "outside" else 0.0)      # (a reference row no star is near)
        e2 = engine.Engine(pack, stars, pr, opt)
        got = e2.logpost(rows[:1], perstar=True)[1]
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin)
        assert np.max(np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin]))) <= 1e-9
        e2.close()
    eng.close()
