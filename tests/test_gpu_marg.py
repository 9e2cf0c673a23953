"""GPU parity of the marginalised mode (k_star_marg, one wavefront per star) against the oracle's
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
