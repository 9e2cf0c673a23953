"""CPU tests of the oracle (oracle/b9_oracle.c): against an independent numpy statement of the
same math, and through the properties SURVEY.md section 8c lists.  BASE-9 parity is UNPINNED
(no reference source, no reference fixtures) -- these tests pin the oracle to the written math.
"""
import numpy as np
import pytest

import oracle
from base_amd import abi, synth
from conftest import build_problem
import numpy_ref


def _oracle(pack, stars, priors, options):
    return oracle.Oracle(pack, stars, priors, options)


@pytest.mark.parametrize("name,n_filt", [("girardi", 3), ("dsed", 8), ("parsec", 8), ("parsec", 5)])
def test_isochrone_matches_numpy(name, n_filt):
    pack_d, cl, pack, stars, priors, options = build_problem(name, n_filt, n_stars=10)
    orc = _oracle(pack, stars, priors, options)
    rng = np.random.default_rng(1)
    for _ in range(20):
        par = synth.default_params(pack_d)
        par[abi.P_LOGAGE] = rng.uniform(pack_d["log_age"][0], pack_d["log_age"][-1])
        par[abi.P_FEH] = rng.uniform(pack_d["feh"][0], pack_d["feh"][-1])
        first, mass, mags, tip = orc.derive_isochrone(par)
        f2, m2, g2 = synth.derive_isochrone(pack_d, par[abi.P_LOGAGE], par[abi.P_FEH], par[abi.P_Y])
        assert first == f2 and len(mass) == len(m2)
        np.testing.assert_allclose(mass, m2, rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(mags, g2, rtol=1e-13, atol=1e-13)
        assert tip == mass[-1]
        assert np.all(np.diff(mass) >= 0)


def test_isochrone_on_grid_node_returns_table_rows():
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=10, ragged=False)
    orc = _oracle(pack, stars, priors, options)
    i_f, i_a = 1, 3
    par = synth.default_params(pack_d)
    par[abi.P_FEH], par[abi.P_LOGAGE] = pack_d["feh"][i_f], pack_d["log_age"][i_a]
    first, mass, mags, tip = orc.derive_isochrone(par)
    k = (i_f * 1 + 0) * len(pack_d["log_age"]) + i_a
    off, n = int(pack_d["iso_offset"][k]), int(pack_d["iso_n_eep"][k])
    assert len(mass) == n
    np.testing.assert_array_equal(mass, pack_d["mass"][off:off + n])
    np.testing.assert_array_equal(mags, pack_d["mags"][off:off + n])


def test_isochrone_outside_grid_is_empty():
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=10)
    orc = _oracle(pack, stars, priors, options)
    for key, val in ((abi.P_LOGAGE, 5.0), (abi.P_LOGAGE, 12.0), (abi.P_FEH, -9.0), (abi.P_FEH, 3.0)):
        par = synth.default_params(pack_d)
        par[key] = val
        first, mass, mags, tip = orc.derive_isochrone(par)
        assert len(mass) == 0
        assert orc.logpost(par[None, :])[0] == -np.inf


def test_ragged_eep_intersection():
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=10, ragged=True)
    orc = _oracle(pack, stars, priors, options)
    par = synth.default_params(pack_d)
    first, mass, mags, tip = orc.derive_isochrone(par)
    la, fe = pack_d["log_age"], pack_d["feh"]
    ia = np.searchsorted(la, par[abi.P_LOGAGE], side="right") - 1
    i_f = np.searchsorted(fe, par[abi.P_FEH], side="right") - 1
    ks = [(i_f + df) * len(la) + ia + da for df in range(2) for da in range(2)]
    lo = max(pack_d["iso_first_eep"][k] for k in ks)
    hi = min(pack_d["iso_first_eep"][k] + pack_d["iso_n_eep"][k] for k in ks)
    assert first == lo and len(mass) == hi - lo


@pytest.mark.parametrize("name,n_filt,wd_frac,n_y,n_pops", [
    ("girardi", 3, 0.0, 1, 1), ("dsed", 8, 0.0, 1, 1), ("parsec", 8, 0.06, 1, 1),
    ("parsec", 8, 0.0, 3, 1), ("parsec", 8, 0.05, 3, 2)])
def test_logpost_matches_numpy(name, n_filt, wd_frac, n_y, n_pops):
    pack_d, cl, pack, stars, priors, options = build_problem(name, n_filt, n_stars=400, wd_frac=wd_frac,
                                                             n_y=n_y, n_pops=n_pops)
    orc = _oracle(pack, stars, priors, options)
    params = synth.walker_params(cl["truth"], 6, n_pops=n_pops)
    lp, ps = orc.logpost(params, perstar=True)
    for w in range(len(params)):
        ref, ref_ps = numpy_ref.logpost(pack_d, cl, priors, params[w], n_pops)
        np.testing.assert_allclose(ps[w], ref_ps, rtol=1e-10, atol=1e-9)
        assert abs(lp[w] - ref) <= 1e-9 * max(1.0, abs(ref))


def test_ifmr_variants_and_wd_types():
    for ifmr in range(6):
        pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=300, wd_frac=0.2, ifmr_id=ifmr)
        orc = _oracle(pack, stars, priors, options)
        par = cl["truth"]
        lp, ps = orc.logpost(par[None, :], perstar=True)
        ref, ref_ps = numpy_ref.logpost(pack_d, cl, priors, par, 1)
        np.testing.assert_allclose(ps[0], ref_ps, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("n_y,n_pops", [(1, 1), (3, 2)])
def test_ragged_wd_cooling_tracks_match_numpy(n_y, n_pops):
    """Cooling tracks with their own age axes (different lengths, ranges, spacings): oracle vs the numpy statement, on a
    cluster rich in WDs of every cooling age -- including ages outside a track's range (clamped bracket, extrapolation)."""
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=500, wd_frac=0.3, n_y=n_y, n_pops=n_pops, wd_ragged=True)
    assert len(set(pack_d["wc_n_age"].tolist())) > 2
    orc = _oracle(pack, stars, priors, options)
    params = synth.walker_params(cl["truth"], 4, n_pops=n_pops)
    params[2, abi.P_LOGAGE] = pack_d["log_age"][0] + 0.02         # a young cluster: short cooling ages
    params[3, abi.P_CARBONICITY] = 0.23                            # near the edge of the carbonicity axis
    lp, ps = orc.logpost(params, perstar=True)
    assert (cl["stage"] == abi.STAGE_WD).sum() > 80
    for w in range(len(params)):
        ref, ref_ps = numpy_ref.logpost(pack_d, cl, priors, params[w], n_pops)
        np.testing.assert_allclose(ps[w], ref_ps, rtol=1e-10, atol=1e-9)
        assert abs(lp[w] - ref) <= 1e-9 * max(1.0, abs(ref))


def test_rectangular_cooling_table_is_a_special_case_of_ragged_tracks():
    """A rectangular table (one shared age axis) given in the ragged form -- every track carrying its own copy of
    the axis -- gives the same bits as the rectangular form abi.make_pack expands."""
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=300, wd_frac=0.3)
    n_tracks, n_age = len(pack_d["wc_carb"]) * len(pack_d["wc_mass"]), len(pack_d["wc_log_age"])
    rag = dict(pack_d, wc_n_age=np.full(n_tracks, n_age, np.int32), wc_offset=np.arange(n_tracks, dtype=np.int64) * n_age,
               wc_log_age=np.tile(pack_d["wc_log_age"], n_tracks))
    params = synth.walker_params(cl["truth"], 3)
    a = _oracle(pack, stars, priors, options).logpost(params, perstar=True)
    b = _oracle(abi.make_pack(rag), stars, priors, options).logpost(params, perstar=True)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_two_pop_lambda_limits_equal_single_pop():
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", 8, n_stars=300, n_y=3, n_pops=2)
    one = _oracle(pack, stars, priors, abi.make_options(n_pops=1))
    two = _oracle(pack, stars, priors, abi.make_options(n_pops=2))
    par = cl["truth"].copy()
    par[abi.P_LAMBDA] = 1.0
    a = one.logpost(par[None, :], perstar=True)[1]
    b = two.logpost(par[None, :], perstar=True)[1]
    np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-12)
    par[abi.P_LAMBDA] = 0.0
    par1 = par.copy()
    par1[abi.P_Y] = par[abi.P_Y2]
    a = one.logpost(par1[None, :], perstar=True)[1]
    b = two.logpost(par[None, :], perstar=True)[1]
    np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-12)


def test_star_order_invariance_and_sigma_monotone():
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=300)
    orc = _oracle(pack, stars, priors, options)
    par = cl["truth"]
    lp, ps = orc.logpost(par[None, :], perstar=True)
    perm = np.random.default_rng(3).permutation(300)
    cl2 = dict(cl)
    for k in ("obs", "sigma", "mass1", "mass_ratio", "clust_prior", "stage", "wd_type"):
        cl2[k] = np.asarray(cl[k])[perm]
    lp2, ps2 = oracle.Oracle(pack, abi.make_stars(cl2), priors, options).logpost(par[None, :], perstar=True)
    np.testing.assert_array_equal(ps2[0], ps[0][perm])
    assert abs(lp2[0] - lp[0]) <= 1e-12 * abs(lp[0])
    # inflating every sigma of a member star far from its prediction raises its likelihood
    cl3 = dict(cl)
    cl3["obs"] = np.array(cl["obs"]) + 1.0
    base = oracle.Oracle(pack, abi.make_stars(cl3), priors, options).logpost(par[None, :], perstar=True)[1][0]
    cl3["sigma"] = np.where(np.array(cl["sigma"]) > 0, np.array(cl["sigma"]) * 5, cl["sigma"])
    wide = oracle.Oracle(pack, abi.make_stars(cl3), priors, options).logpost(par[None, :], perstar=True)[1][0]
    assert np.all(wide >= base - 1e-12)


def test_wd_branch_reached_and_finite():
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=400, wd_frac=0.25)
    orc = _oracle(pack, stars, priors, options)
    lp, ps = orc.logpost(cl["truth"][None, :], perstar=True)
    wd = cl["stage"] == abi.STAGE_WD
    assert wd.sum() > 50 and np.all(np.isfinite(ps[0]))
    # WD members drawn from the model must be likelier under the cluster than under the field alone
    log_fs = -np.sum(np.log(cl["filter_prior_max"] - cl["filter_prior_min"]))
    member = wd & ~cl["is_field"]
    assert np.mean(ps[0][member] > np.log1p(-cl["clust_prior"][member]) + log_fs + 1.0) > 0.9


def test_priors():
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 3, n_stars=20)
    lib = oracle.load()
    import ctypes as C
    par = cl["truth"].copy()
    f = lambda p: lib.b9o_log_prior_cluster(C.byref(priors), p.ctypes.data_as(C.POINTER(C.c_double)), 1)
    assert f(par) == 0.0
    p2 = par.copy(); p2[abi.P_FEH] += 0.3
    assert abs(f(p2) + 0.5) < 1e-12
    p3 = par.copy(); p3[abi.P_ABS] = -0.01
    assert f(p3) == -np.inf
    p4 = par.copy(); p4[abi.P_LOGAGE] = pack_d["log_age"][-1] + 0.01
    assert f(p4) == -np.inf
    # IMF normalisation integrates to one over [0.1, m_wd_up]
    m = np.linspace(0.1, 8.0, 400001)
    dens = np.exp([lib.b9o_log_prior_mass(lib.b9o_log_mass_norm(8.0), float(x)) for x in m[::400]])
    from scipy.integrate import simpson
    assert abs(simpson(dens, x=m[::400]) - 1.0) < 2e-3


def test_marginalised_mode_matches_numpy_brute_force():
    """Third opinion on the marginalised integral (MS/RGB-stage stars): numpy brute force vs the oracle."""
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", 5, n_stars=12, n_feh=3, n_age=4, n_eep=25)
    K, Q = 2, 3
    orc = oracle.Oracle(pack, stars, priors, abi.make_options(abi.MODE_MARGINALISED, 1, K, Q))
    par = cl["truth"]
    lp, ps = orc.logpost(par[None, :], perstar=True)
    ll = numpy_ref.marg_perstar(pack_d, cl, par, K, Q)
    log_fs = -np.sum(np.log(cl["filter_prior_max"] - cl["filter_prior_min"]))
    pm = cl["clust_prior"]
    from scipy.special import logsumexp
    want = logsumexp(np.stack([np.log1p(-pm) + log_fs, np.log(pm) + ll]), axis=0)
    np.testing.assert_allclose(ps[0], want, rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize("n_pops,n_y,wd_frac,K,Q", [(1, 1, 0.25, 2, 3), (2, 3, 0.0, 2, 2), (2, 3, 0.2, 1, 3)])
def test_marginalised_wd_stage_and_two_populations_match_numpy_brute_force(n_pops, n_y, wd_frac, K, Q):
    """Third statement of the two pieces of the marginalised mode that had only two (oracle and kernel, one author): the
    WD-stage stars' integral over (AGB tip, M_wd_up] through the WD branch (DA and DB), and the two-population mixture of
    marginals -- numpy brute force over synth.forward_mags against the oracle, every star, 1e-10."""
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", 5, n_stars=16, wd_frac=wd_frac, n_y=n_y, n_pops=n_pops,
                                                       n_feh=3, n_age=4, n_eep=25, seed=12)
    stage = np.asarray(cl["stage"])
    if wd_frac > 0:     # both atmosphere types among the WD-stage stars
        idx = np.nonzero(stage == abi.STAGE_WD)[0]
        assert len(idx) >= 3
        cl["wd_type"][idx[::2]] = 1
        cl["wd_type"][idx[1::2]] = 0
        stars = abi.make_stars(cl)
    orc = oracle.Oracle(pack, stars, priors, abi.make_options(abi.MODE_MARGINALISED, n_pops, K, Q))
    rows = synth.walker_params(cl["truth"], 2, seed=4, n_pops=n_pops, scale=0.05)
    lp, ps = orc.logpost(rows, perstar=True)
    for r in range(2):
        want_lp, want = numpy_ref.marg_logpost(pack_d, cl, priors, rows[r], K, Q, n_pops)
        assert np.all(np.isfinite(want))
        np.testing.assert_allclose(ps[r], want, rtol=1e-10, atol=1e-10)
        assert abs(lp[r] - want_lp) <= 1e-10 * max(1.0, abs(want_lp))


def test_sample_mass_restatement_properties():
    """CPU-side pins of the sampleMass restatement: its Philox equals the numpy twin (itself pinned to the
    Random123 vectors in test_mcmc.py), draws are grid nodes, membership is what the marginal implies,
    and a row's draws depend on (seed, row index) only."""
    import ctypes as C
    from base_amd import mcmc
    lib = oracle.load()
    for ctr, key in (((0, 0, 0, 0), (0, 0)), ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2), ((1, 2, 3, 4), (5, 6))):
        out = (C.c_uint32 * 4)()
        lib.b9o_philox4x32((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        tw = mcmc.philox4x32(*(np.uint32(x) for x in ctr), *(np.uint32(x) for x in key))
        assert [int(x) for x in out] == [int(x) for x in tw]
    pack_d, cl, pack, stars, priors, _ = build_problem("parsec", 4, n_stars=40, wd_frac=0.1, seed=3)
    opt = abi.make_options(mode=abi.MODE_MARGINALISED, marg_iso_increm=2, marg_n_q=3)
    o = oracle.Oracle(pack, stars, priors, opt)
    rows = np.tile(cl["truth"], (6, 1))
    m, q, mem, pop, margin = o.sample_mass(rows, seed=7, row0=10)
    assert set(np.unique(q)) <= {0.0, 1.0 / 3.0, 2.0 / 3.0} and np.all(pop == 0) and np.all(margin > 0)
    assert np.all(m > 0) and np.all(m <= pack_d["m_wd_up"])
    _, ps = o.logpost(rows[:1], perstar=True)
    log_fs = -np.sum(np.log(cl["filter_prior_max"] - cl["filter_prior_min"]))
    want = 1.0 - np.exp(np.log1p(-np.asarray(cl["clust_prior"])) + log_fs - ps[0])
    np.testing.assert_allclose(mem[0], np.clip(want, 0.0, 1.0), rtol=1e-9, atol=1e-12)
    assert not np.array_equal(m[0], m[1])                                # same parameters, other row index -> other draws
    m2 = o.sample_mass(rows[3:], seed=7, row0=13)[0]
    assert np.array_equal(m[3:], m2)
