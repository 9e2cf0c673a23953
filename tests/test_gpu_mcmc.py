"""GPU tests of the device-resident Metropolis block (b9_mcmc_run_block) against the host
reference runner (base_amd.mcmc.HostBlockRunner) driving the same GPU log-posterior: same
counter-based random numbers, so the chains agree step for step (to fp64 rounding of the
Box-Muller transcendentals) and the accept counts are equal."""
import numpy as np
import pytest

import oracle
from base_amd import abi, mcmc, synth
from conftest import build_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_pops,free", [(1, mcmc.DEFAULT_FREE),
                                          (2, (abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS, abi.P_Y, abi.P_Y2, abi.P_LAMBDA))])
def test_device_block_matches_host_reference(n_pops, free):
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=600, wd_frac=0.03,
                                                             n_y=3 if n_pops == 2 else 1, n_pops=n_pops)
    eng = engine.Engine(pack, stars, priors, options)
    start = synth.walker_params(cl["truth"], 6, seed=42, scale=0.1, n_pops=n_pops)
    ids = np.arange(10, 16)
    d = len(free)
    rng = np.random.default_rng(0)
    chol = np.tril(rng.normal(size=(d, d))) * 2e-4 + np.diag(np.full(d, 1e-3))
    lp0 = eng.logpost(start)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, ids, np.array(free), chol, 77, 1000, 40)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, ids, np.array(free), chol, 77, 1000, 40)
    assert dev[4] == host[4] and 0 < dev[4] < 40 * 6
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)     # samples
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)                  # log-posteriors
    np.testing.assert_allclose(dev[0], host[0], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[1], host[1], rtol=1e-10)
    # and the chain's log-posteriors are what the oracle says for those positions
    orc = oracle.Oracle(pack, stars, priors, options)
    want = orc.logpost(dev[0])
    assert np.max(np.abs(dev[1] - want) / np.maximum(1.0, np.abs(want))) <= 1e-9


def test_walker_sampler_on_device_recovers_truth():
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=2000, small=False)
    eng = engine.Engine(pack, stars, priors, options)
    start = synth.walker_params(cl["truth"], 8, seed=1, scale=0.05)
    s = mcmc.WalkerSampler(start, mcmc.DeviceBlockRunner(eng), block=50, seed=5)
    s.initialise(eng.logpost)
    rec = []
    s.run(1000, rec)
    samples = np.concatenate([r[0] for r in rec])[500:]          # [steps, walkers, d]
    mean, sd = samples.mean(axis=(0, 1)), samples.std(axis=(0, 1))
    truth = cl["truth"][list(mcmc.DEFAULT_FREE)]
    assert 0.05 < s.accepted / (1000 * 8) < 0.8
    assert np.all(np.abs(mean - truth) < 6 * sd + 1e-3), (mean, truth, sd)
