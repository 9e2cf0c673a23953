"""Property-based GPU parity (hypothesis): random in-grid (and just-outside) cluster parameters on the three
synthetic pack families, both evaluation modes, one and two populations -- the HIP path against the oracle,
per star, to the stated fp64 tolerance; and the device-resident sampler against the host twin from random states."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
from base_amd import abi, mcmc, synth
from conftest import build_problem

pytestmark = pytest.mark.gpu

_CACHE = {}


def _setup(key):
    if key not in _CACHE:
        from base_amd import engine
        name, nf, ny, npops, mode = key
        pack_d, cl, pack, stars, priors, _ = build_problem(name, nf, n_stars=260, wd_frac=0.08, n_y=ny, n_pops=npops, seed=31)
        opt = abi.make_options(mode=mode, n_pops=npops, marg_iso_increm=2, marg_n_q=3)
        _CACHE[key] = (pack_d, cl, engine.Engine(pack, stars, priors, opt), oracle.Oracle(pack, stars, priors, opt))
    return _CACHE[key]


KEYS = [("girardi", 3, 1, 1, abi.MODE_GIVEN_MASS), ("parsec", 8, 1, 1, abi.MODE_GIVEN_MASS), ("dsed", 5, 3, 2, abi.MODE_GIVEN_MASS),
        ("parsec", 4, 1, 1, abi.MODE_MARGINALISED), ("dsed", 5, 3, 2, abi.MODE_MARGINALISED)]
unit = st.floats(min_value=-0.02, max_value=1.02, allow_nan=False)          # a little beyond both grid edges


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(key=st.sampled_from(KEYS), ua=unit, uf=unit, uy=unit, uy2=unit, lam=st.floats(0.0, 1.0), dmod=st.floats(-0.5, 0.5),
       av=st.floats(-0.01, 0.5), carb=st.floats(0.0, 1.0), ifmr=st.floats(-0.05, 0.05))
def test_logpost_matches_oracle_anywhere(key, ua, uf, uy, uy2, lam, dmod, av, carb, ifmr):
    pack_d, cl, eng, orc = _setup(key)
    par = cl["truth"].copy()
    la, fe, yy = pack_d["log_age"], pack_d["feh"], pack_d["y"]
    par[abi.P_LOGAGE] = la[0] + ua * (la[-1] - la[0])
    par[abi.P_FEH] = fe[0] + uf * (fe[-1] - fe[0])
    par[abi.P_Y] = yy[0] + uy * (yy[-1] - yy[0])
    par[abi.P_Y2] = yy[0] + uy2 * (yy[-1] - yy[0])
    par[abi.P_LAMBDA], par[abi.P_ABS], par[abi.P_CARBONICITY] = lam, av, carb
    par[abi.P_MOD] += dmod
    par[abi.P_IFMR_SLOPE] += ifmr
    got, got_ps = eng.logpost(par[None, :], perstar=True)
    want, want_ps = orc.logpost(par[None, :], perstar=True)
    fin = np.isfinite(want_ps)
    assert np.array_equal(np.isfinite(got_ps), fin)
    if fin.any():
        assert np.max(np.abs(got_ps[fin] - want_ps[fin]) / np.maximum(1.0, np.abs(want_ps[fin]))) <= 1e-9
    assert (got[0] == want[0]) or abs(got[0] - want[0]) <= 1e-9 * max(1.0, abs(want[0]))


@settings(max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 2**31 - 1), step0=st.integers(0, 2**40), n_steps=st.integers(1, 12), n_w=st.integers(1, 9),
       scale=st.floats(0.2, 3.0))
def test_device_block_matches_host_twin_from_random_states(seed, step0, n_steps, n_w, scale):
    pack_d, cl, eng, orc = _setup(("parsec", 8, 1, 1, abi.MODE_GIVEN_MASS))
    free = np.array(mcmc.DEFAULT_FREE)
    start = synth.walker_params(cl["truth"], n_w, seed=seed % 1000, scale=0.1)
    chol = np.diag([3e-4, 2e-3, 8e-4, 6e-4]) * scale
    ids = (np.arange(n_w) * 7 + seed % 5).astype(np.int32)
    lp0 = eng.logpost(start)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, ids, free, chol, seed, step0, n_steps)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, ids, free, chol, seed, step0, n_steps)
    assert dev[4] == host[4]
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)


def _ragged_pack(rng, n_filt, with_wd):
    """A small random model pack with RAGGED isochrones: every (FeH, Y, age) isochrone has its own first EEP and
    length, so corner EEP ranges intersect partially, barely (2 points) or not at all."""
    base = synth.make_pack("girardi", n_filt, n_feh=2, n_age=2, n_eep=8)          # for the WD tables and coefficients
    n_feh, n_y, n_age = int(rng.integers(2, 4)), int(rng.integers(1, 3)), int(rng.integers(2, 4))
    feh = np.sort(rng.uniform(-2, 0.5, n_feh)) + np.arange(n_feh) * 0.05
    y = np.sort(rng.uniform(0.23, 0.33, n_y)) + np.arange(n_y) * 0.01
    log_age = np.sort(rng.uniform(8.0, 10.0, n_age)) + np.arange(n_age) * 0.05
    first, cnt, off, mass, mags = [], [], [], [], []
    for _ in range(n_feh * n_y * n_age):
        f0, n = int(rng.integers(0, 4)), int(rng.integers(2, 14))
        m = np.cumsum(rng.uniform(0.0, 0.4, n)) + rng.uniform(0.1, 0.3)           # non-descending, sometimes flat
        first.append(f0); cnt.append(n); off.append(len(mass))
        mass.extend(m)
        mags.extend((12.0 - 6.0 * np.log10(m)[:, None] + rng.normal(0, 0.3, (n, n_filt)) + np.arange(n_filt) * 0.2).ravel())
    d = dict(base)
    d.update(feh=feh, y=y, log_age=log_age, iso_first_eep=np.array(first), iso_n_eep=np.array(cnt), iso_offset=np.array(off),
             mass=np.array(mass), mags=np.array(mags), n_filt=n_filt)
    if not with_wd:
        for k in ("wc_carb", "wc_mass", "wc_log_age", "wc_log_teff", "wc_log_radius", "at_logg", "at_log_teff", "at_mags"):
            d[k] = np.zeros(0)
        d["n_at_type"] = 0
    return d


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 10**6), n_filt=st.sampled_from([1, 3, 5, 8]), with_wd=st.booleans(), two_pops=st.booleans(),
       marg=st.booleans())
def test_ragged_random_packs_match_oracle(seed, n_filt, with_wd, two_pops, marg):
    from base_amd import engine
    rng = np.random.default_rng(seed)
    pack_d = _ragged_pack(rng, n_filt, with_wd)
    n_pops = 2 if (two_pops and len(pack_d["y"]) > 1) else 1
    n = 150
    sig = rng.uniform(0.02, 0.2, (n, n_filt)); sig[rng.random((n, n_filt)) < 0.15] = -1.0
    cl = dict(n_filt=n_filt, obs=rng.uniform(8, 16, (n, n_filt)).ravel(), sigma=sig.ravel(), mass1=rng.uniform(0.05, 5.0, n),
              mass_ratio=np.where(rng.random(n) < 0.4, rng.uniform(0, 1, n), 0.0), clust_prior=rng.uniform(0.05, 1.0, n),
              stage=np.where(rng.random(n) < 0.1, abi.STAGE_WD, abi.STAGE_MSRG).astype(np.int32), wd_type=(rng.random(n) < 0.3).astype(np.int32),
              filter_prior_min=np.full(n_filt, 7.0), filter_prior_max=np.full(n_filt, 17.0))
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors = abi.make_priors(log_age_min=-1e9, log_age_max=1e9)
    opt = abi.make_options(mode=abi.MODE_MARGINALISED if marg else abi.MODE_GIVEN_MASS, n_pops=n_pops, marg_iso_increm=2, marg_n_q=3)
    rows = []
    for _ in range(6):
        r = np.zeros(abi.B9_NPARAM)
        r[abi.P_LOGAGE] = rng.uniform(pack_d["log_age"][0] - 0.05, pack_d["log_age"][-1] + 0.05)
        r[abi.P_FEH] = rng.uniform(pack_d["feh"][0] - 0.05, pack_d["feh"][-1] + 0.05)
        r[abi.P_Y], r[abi.P_Y2] = rng.uniform(pack_d["y"][0], pack_d["y"][-1], 2)
        r[abi.P_MOD], r[abi.P_ABS], r[abi.P_CARBONICITY] = rng.uniform(-1, 1), rng.uniform(0, 0.5), rng.uniform(0, 1)
        r[abi.P_IFMR_INTERCEPT], r[abi.P_IFMR_SLOPE], r[abi.P_LAMBDA] = 0.77, 0.08, rng.uniform(0.05, 0.95)
        rows.append(r)
    rows = np.array(rows)
    got, got_ps = engine.Engine(pack, stars, priors, opt).logpost(rows, perstar=True)
    want, want_ps = oracle.Oracle(pack, stars, priors, opt).logpost(rows, perstar=True)
    fin = np.isfinite(want_ps)
    assert np.array_equal(np.isfinite(got_ps), fin) and not np.isnan(got_ps).any()
    if fin.any():
        assert np.max(np.abs(got_ps[fin] - want_ps[fin]) / np.maximum(1.0, np.abs(want_ps[fin]))) <= 1e-9
    f = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), f)
    if f.any():
        assert np.max(np.abs(got[f] - want[f]) / np.maximum(1.0, np.abs(want[f]))) <= 1e-9
    if marg:       # the sampleMass draws on the same ragged grid
        gm, gq, gmem, gpop = engine.Engine(pack, stars, priors, opt).sample_mass(rows[:3], seed=seed, row0=3)
        om, oq, omem, opop, margin = oracle.Oracle(pack, stars, priors, opt).sample_mass(rows[:3], seed=seed, row0=3)
        safe = margin > 1e-6
        assert np.array_equal(gq[safe], oq[safe]) and np.array_equal(gpop[safe], opop[safe])
        np.testing.assert_allclose(gm[safe], om[safe], rtol=1e-12, atol=0)
        np.testing.assert_allclose(gmem, omem, rtol=1e-8, atol=1e-300)


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 10**6), n_filt=st.sampled_from([3, 8]), two_pops=st.booleans(), scale=st.floats(0.3, 30.0))
def test_sampler_on_ragged_packs_matches_host_twin(seed, n_filt, two_pops, scale):
    """Large steps on a ragged grid: proposals keep falling where the corner isochrones share fewer than two EEPs
    or outside the grid altogether -- invalid candidates must be rejected exactly as the host twin rejects them."""
    from base_amd import engine
    rng = np.random.default_rng(seed)
    pack_d = _ragged_pack(rng, n_filt, True)
    n_pops = 2 if (two_pops and len(pack_d["y"]) > 1) else 1
    n = 200
    cl = dict(n_filt=n_filt, obs=rng.uniform(8, 16, (n, n_filt)).ravel(), sigma=rng.uniform(0.05, 0.3, (n, n_filt)).ravel(),
              mass1=rng.uniform(0.05, 5.0, n), mass_ratio=np.where(rng.random(n) < 0.4, rng.uniform(0, 1, n), 0.0),
              clust_prior=rng.uniform(0.05, 1.0, n), stage=np.where(rng.random(n) < 0.1, abi.STAGE_WD, abi.STAGE_MSRG).astype(np.int32),
              wd_type=(rng.random(n) < 0.3).astype(np.int32), filter_prior_min=np.full(n_filt, 7.0), filter_prior_max=np.full(n_filt, 17.0))
    eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), abi.make_priors(log_age_min=-1e9, log_age_max=1e9), abi.make_options(n_pops=n_pops))
    W = 5
    start = np.zeros((W, abi.B9_NPARAM))
    start[:, abi.P_LOGAGE] = rng.uniform(pack_d["log_age"][0], pack_d["log_age"][-1], W)
    start[:, abi.P_FEH] = rng.uniform(pack_d["feh"][0], pack_d["feh"][-1], W)
    start[:, abi.P_Y] = rng.uniform(pack_d["y"][0], pack_d["y"][-1], W); start[:, abi.P_Y2] = rng.uniform(pack_d["y"][0], pack_d["y"][-1], W)
    start[:, abi.P_MOD], start[:, abi.P_ABS], start[:, abi.P_LAMBDA] = 0.2, 0.1, 0.5
    start[:, abi.P_IFMR_INTERCEPT], start[:, abi.P_IFMR_SLOPE] = 0.77, 0.08
    free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS] + ([abi.P_Y, abi.P_Y2, abi.P_LAMBDA] if n_pops == 2 else []))
    chol = np.diag([0.05, 0.1, 0.02, 0.02] + ([0.01, 0.01, 0.05] if n_pops == 2 else [])) * scale
    lp0 = eng.logpost(start)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, np.arange(W), free, chol, seed, 7, 14)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, np.arange(W), free, chol, seed, 7, 14)
    assert dev[4] == host[4]
    np.testing.assert_array_equal(np.isfinite(dev[3]), np.isfinite(host[3]))
    fin = np.isfinite(host[3])
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3][fin], host[3][fin], rtol=1e-10)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 10**6), n_filt=st.sampled_from([1, 4, 8, 12]))
def test_derived_isochrone_is_bit_exact_on_ragged_packs(seed, n_filt):
    """b9_derive_isochrone (what makeCMD prints and every step uses) against the oracle, bit for bit, where the corner
    isochrones' EEP ranges overlap fully, partly, by two points, or not at all."""
    from base_amd import engine
    rng = np.random.default_rng(seed)
    pack_d = _ragged_pack(rng, n_filt, False)
    pack = abi.make_pack(pack_d)
    cl = dict(n_filt=n_filt, obs=np.full(n_filt, 10.0), sigma=np.full(n_filt, 0.1), mass1=np.array([1.0]), mass_ratio=np.array([0.0]),
              clust_prior=np.array([0.9]), filter_prior_min=np.full(n_filt, 7.0), filter_prior_max=np.full(n_filt, 17.0))
    eng = engine.Engine(pack, abi.make_stars(cl), abi.make_priors(), abi.make_options())
    lib = oracle.load()
    for _ in range(8):
        r = np.zeros(abi.B9_NPARAM)
        r[abi.P_LOGAGE] = rng.uniform(pack_d["log_age"][0] - 0.02, pack_d["log_age"][-1] + 0.02)
        r[abi.P_FEH] = rng.uniform(pack_d["feh"][0] - 0.02, pack_d["feh"][-1] + 0.02)
        r[abi.P_Y] = rng.uniform(pack_d["y"][0], pack_d["y"][-1]); r[abi.P_Y2] = rng.uniform(pack_d["y"][0], pack_d["y"][-1])
        for pop in (0, 1):
            g = eng.derive_isochrone(r, pop=pop, cap=64)
            w = oracle.derive_isochrone(lib, pack, r, pop=pop, cap=64)
            assert g[0] == w[0] and g[3] == w[3]
            np.testing.assert_array_equal(g[1], w[1])
            np.testing.assert_array_equal(g[2], w[2])


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(seed=st.integers(0, 10**6), log_sig=st.floats(-6.0, 3.0), log_field=st.floats(-300.0, 2.0), marg=st.booleans())
def test_extreme_likelihood_scales(seed, log_sig, log_field, marg):
    """sigmas from 1e-6 to 1e3 mag (chi^2 up to ~1e14: exp underflow, the additive branch of the product-form
    mixture when e^l would overflow) and field-star densities down to 1e-300: same values as the oracle."""
    pack_d, cl0, _, _ = _setup(("parsec", 8, 1, 1, abi.MODE_GIVEN_MASS))
    from base_amd import engine
    rng = np.random.default_rng(seed)
    cl = dict(cl0)
    n = len(cl0["mass1"])
    cl["sigma"] = (10.0 ** log_sig * rng.uniform(0.5, 2.0, (n, 8))).ravel()
    width = 10.0 ** (-log_field / 8.0)                                    # box volume = width^8 -> field-star density 10^log_field
    cl["filter_prior_min"], cl["filter_prior_max"] = np.full(8, 10.0), np.full(8, 10.0 + width)
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors = synth.default_priors(pack_d, cl0["truth"])
    opt = abi.make_options(mode=abi.MODE_MARGINALISED if marg else abi.MODE_GIVEN_MASS, marg_iso_increm=2, marg_n_q=2)
    rows = synth.walker_params(cl0["truth"], 3, seed=seed % 97, scale=0.5)
    got, got_ps = engine.Engine(pack, stars, priors, opt).logpost(rows, perstar=True)
    want, want_ps = oracle.Oracle(pack, stars, priors, opt).logpost(rows, perstar=True)
    assert not np.isnan(got_ps).any()
    fin = np.isfinite(want_ps)
    assert np.array_equal(np.isfinite(got_ps), fin)
    assert np.max(np.abs(got_ps[fin] - want_ps[fin]) / np.maximum(1.0, np.abs(want_ps[fin]))) <= 1e-9
    f = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), f)
    if f.any():
        assert np.max(np.abs(got[f] - want[f]) / np.maximum(1.0, np.abs(want[f]))) <= 1e-9


@settings(max_examples=20, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(sizes=st.lists(st.integers(1, 9), min_size=1, max_size=7), record=st.lists(st.booleans(), min_size=7, max_size=7),
       depth=st.integers(1, 2), seed=st.integers(0, 10**6), n_w=st.integers(1, 6))
def test_block_pipeline_any_schedule(sizes, record, depth, seed, n_w):
    """B9_BLOCK_CONTINUE | B9_BLOCK_ASYNC under random block lengths, record on/off and one or two outstanding blocks:
    the chain of the synchronous block-by-block calls."""
    pack_d, cl, eng, orc = _setup(("parsec", 8, 1, 1, abi.MODE_GIVEN_MASS))
    free, chol = np.array(mcmc.DEFAULT_FREE), np.diag([3e-4, 2e-3, 8e-4, 6e-4])
    start = synth.walker_params(cl["truth"], n_w, seed=seed % 100, scale=0.1)
    lp0 = eng.logpost(start)
    ids = np.arange(n_w)
    p, lp, s0, want = start, lp0, 77, []
    for k, n in enumerate(sizes):
        p, lp, s, l, a = eng.mcmc_run_block(p, lp, ids, free, chol * (1 + 0.05 * k), seed, s0, n, record=record[k])
        want.append((s, l, a)); s0 += n
    hs, s0, got = [], 77, []
    for k, n in enumerate(sizes):
        hs.append(eng.mcmc_submit(start, lp0, ids, free, chol * (1 + 0.05 * k), seed, s0, n, record[k], cont=k > 0))
        s0 += n
        while len(hs) >= (2 if depth == 2 else 1):
            r = eng.mcmc_collect(hs.pop(0)); got.append((r[2], r[3], r[4]))
    while hs:
        r = eng.mcmc_collect(hs.pop(0)); got.append((r[2], r[3], r[4]))
    for (ws, wl, wa), (gs, gl, ga) in zip(want, got):
        assert wa == ga
        assert (ws is None) == (gs is None)
        if ws is not None:
            np.testing.assert_array_equal(ws, gs); np.testing.assert_array_equal(wl, gl)
    np.testing.assert_array_equal(r[0], p); np.testing.assert_array_equal(r[1], lp)
