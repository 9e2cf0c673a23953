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
