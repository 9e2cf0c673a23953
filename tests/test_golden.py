"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py).

CPU: the oracle reproduces them bit for bit.  GPU: the HIP path reproduces the isochrone bit for
bit and the log-posteriors to the stated fp64 tolerance.  The fixtures are this repo's own
(BASE-9 parity unpinned -- see make_golden.py)."""
import numpy as np
import pytest

import golden_util
import oracle
from base_amd import abi, mcmc


@pytest.mark.parametrize("name", golden_util.names())
def test_oracle_reproduces_golden(name):
    z, pack_d, cl, pack, stars, priors, options = golden_util.load(name)
    lp, ps = oracle.Oracle(pack, stars, priors, options).logpost(z["params"], perstar=True)
    np.testing.assert_array_equal(lp, z["logpost"])
    np.testing.assert_array_equal(ps, z["perstar"])
    iso = oracle.derive_isochrone(oracle.load(), pack, z["params"][0])
    assert iso[0] == int(z["iso_first"]) and iso[3] == float(z["iso_tip"])
    np.testing.assert_array_equal(iso[1], z["iso_mass"])
    np.testing.assert_array_equal(iso[2], z["iso_mags"])


def _marg_options(options):
    return abi.make_options(mode=abi.MODE_MARGINALISED, n_pops=options.n_pops, marg_iso_increm=2, marg_n_q=3)


@pytest.mark.parametrize("name", golden_util.names())
def test_oracle_reproduces_golden_marginalised_draws_and_chain(name):
    z, pack_d, cl, pack, stars, priors, options = golden_util.load(name)
    orc_m = oracle.Oracle(pack, stars, priors, _marg_options(options))
    lp, ps = orc_m.logpost(z["params"], perstar=True)
    np.testing.assert_array_equal(lp, z["marg_logpost"])
    np.testing.assert_array_equal(ps, z["marg_perstar"])
    sm = orc_m.sample_mass(z["params"][:3], seed=11, row0=5)
    for got, key in zip(sm, ("sm_mass", "sm_ratio", "sm_member", "sm_pop", "sm_margin")):
        np.testing.assert_array_equal(got, z[key])
    orc = oracle.Oracle(pack, stars, priors, options)
    start = z["params"][:3].copy()
    chain = mcmc.HostBlockRunner(orc.logpost).run(start, orc.logpost(start), np.array([0, 1, 2]), z["chain_free"], z["chain_chol"], 2024, 40, 20)
    np.testing.assert_array_equal(chain[2], z["chain_samples"])
    np.testing.assert_array_equal(chain[3], z["chain_lps"])
    assert chain[4] == int(z["chain_accepted"]) and 0 < chain[4] < 60


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_util.names())
def test_hip_reproduces_golden_marginalised_draws_and_chain(name):
    from base_amd import engine
    z, pack_d, cl, pack, stars, priors, options = golden_util.load(name)
    eng_m = engine.Engine(pack, stars, priors, _marg_options(options))
    lp, ps = eng_m.logpost(z["params"], perstar=True)
    fin = np.isfinite(z["marg_perstar"])
    assert np.array_equal(np.isfinite(ps), fin)
    assert np.max(np.abs(ps[fin] - z["marg_perstar"][fin]) / np.maximum(1.0, np.abs(z["marg_perstar"][fin]))) <= 1e-9
    f = np.isfinite(z["marg_logpost"])
    assert np.array_equal(np.isfinite(lp), f)
    assert np.max(np.abs(lp[f] - z["marg_logpost"][f]) / np.maximum(1.0, np.abs(z["marg_logpost"][f]))) <= 1e-9
    m, q, mem, pop = eng_m.sample_mass(z["params"][:3], seed=11, row0=5)
    safe = z["sm_margin"] > 1e-6
    assert safe.mean() > 0.999
    np.testing.assert_array_equal(q[safe], z["sm_ratio"][safe])
    np.testing.assert_array_equal(pop[safe], z["sm_pop"][safe])
    np.testing.assert_allclose(m[safe], z["sm_mass"][safe], rtol=1e-12, atol=0)
    np.testing.assert_allclose(mem, z["sm_member"], rtol=1e-9, atol=1e-300)
    # the device-resident sampler (fused one-launch step) reproduces the golden chain
    eng = engine.Engine(pack, stars, priors, options)
    start = z["params"][:3].copy()
    dev = mcmc.DeviceBlockRunner(eng).run(start, eng.logpost(start), np.array([0, 1, 2]), z["chain_free"], z["chain_chol"], 2024, 40, 20)
    assert dev[4] == int(z["chain_accepted"])
    np.testing.assert_allclose(dev[2], z["chain_samples"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(dev[3], z["chain_lps"], rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_util.names())
def test_hip_reproduces_golden(name):
    from base_amd import engine
    z, pack_d, cl, pack, stars, priors, options = golden_util.load(name)
    eng = engine.Engine(pack, stars, priors, options)
    lp, ps = eng.logpost(z["params"], perstar=True)
    want_lp, want_ps = z["logpost"], z["perstar"]
    fin = np.isfinite(want_ps)
    assert np.array_equal(np.isfinite(ps), fin)
    assert np.max(np.abs(ps[fin] - want_ps[fin]) / np.maximum(1.0, np.abs(want_ps[fin]))) <= 1e-9
    f = np.isfinite(want_lp)
    assert np.array_equal(np.isfinite(lp), f) and np.all(lp[~f] == want_lp[~f])
    assert np.max(np.abs(lp[f] - want_lp[f]) / np.maximum(1.0, np.abs(want_lp[f]))) <= 1e-9
    iso = eng.derive_isochrone(z["params"][0])
    assert iso[0] == int(z["iso_first"]) and iso[3] == float(z["iso_tip"])
    np.testing.assert_array_equal(iso[1], z["iso_mass"])
    np.testing.assert_array_equal(iso[2], z["iso_mags"])
