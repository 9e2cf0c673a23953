"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py).

CPU: the oracle reproduces them bit for bit.  GPU: the HIP path reproduces the isochrone bit for
bit and the log-posteriors to the stated fp64 tolerance.  The fixtures are this repo's own
(BASE-9 parity unpinned -- see make_golden.py)."""
import numpy as np
import pytest

import golden_util
import oracle


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
