"""CPU tests of the C++ host driver (base_amd/host/b9sampler.cpp through libbase9host.so): the walker-parallel adaptive
sampler's own logic -- block schedule, one-block adaptation lag, pooled moments, sharding -- with the block runner and
the all-gather supplied by the test (the seam of include/base9_host.h: the oracle evaluates, gloo gathers).  With
world_size 2 over gloo every chain is bit-identical to the one-rank run.  The product path (GPU runner, RCCL exchange)
is covered by the `gpu` tests in test_gpu_sampler.py."""
import os
import socket

import numpy as np
import pytest

import oracle
from base_amd import abi, host_build, hostlib, mcmc, synth
from conftest import build_problem

FREE = (abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS)
STEP = [mcmc.DEFAULT_STEP[k] for k in FREE]


@pytest.fixture(scope="module", autouse=True)
def _built():
    from base_amd import build
    build.build_hip()
    host_build.build_host()


def _problem():
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=150, seed=11)
    return pack_d, cl, oracle.Oracle(pack, stars, priors, options)


def _run(rank, world, exchange, n_walkers=4, burn=60, main=20, block=20):
    pack_d, cl, orc = _problem()
    start = synth.walker_params(cl["truth"], n_walkers, seed=42, scale=0.2)
    twin = mcmc.HostBlockRunner(orc.logpost)
    s = hostlib.HostSampler(n_walkers, FREE, STEP, exchange, seed=99, block=block, run_block=twin.run, evaluate=orc.logpost)
    s.initialise(start)
    a = s.run(burn, adapt=True, record=True)
    b = s.run(main, adapt=False, record=True)
    return s, np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]])


def test_summary_rows_match_numpy():
    rng = np.random.default_rng(3)
    n, wl, d = 37, 5, 4
    samples = rng.normal(size=(n, wl, d)) * 1e-3 + np.array([9.1, 0.0, 10.2, 0.3])
    samples[5] = samples[4]                     # a step on which nobody moved
    samples[9, 2] = samples[8, 2]
    params_end = rng.normal(size=(wl, abi.B9_NPARAM))
    lp_end = rng.normal(size=wl)
    origin = samples[0].mean(axis=0)
    rows = hostlib.summary_rows(samples, params_end, lp_end, origin)
    assert rows.shape == (wl, abi.row_doubles(d))
    x = samples - origin
    np.testing.assert_array_equal(rows[:, 0], lp_end)
    np.testing.assert_array_equal(rows[:, 1:13], params_end)
    np.testing.assert_array_equal(rows[:, 13], (np.abs(np.diff(samples, axis=0)).sum(axis=2) > 0).sum(axis=0))
    np.testing.assert_array_equal(rows[:, 14], n)
    np.testing.assert_allclose(rows[:, 15:15 + d], x.sum(axis=0), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(rows[:, 15 + d:].reshape(wl, d, d), np.einsum("swi,swj->wij", x, x), rtol=1e-12, atol=1e-18)


def test_sampler_moves_adapts_and_freezes():
    s, samples, lps = _run(0, 1, hostlib.Exchange.local(), burn=200, main=50, block=25)
    st = s.state()
    assert st["steps"] == 250 and 0.05 < st["accepted_local"] / (250 * 4) < 0.95
    assert not np.allclose(st["chol"], np.diag(np.diag(st["chol"])))           # adapted: no longer diagonal
    assert np.all(np.isfinite(st["all_logpost"]))
    assert lps[-50:].mean() >= lps[:50].mean() - 1.0                           # not drifting away from the mode
    # the main run leaves the proposal alone
    before = (st["scale"], st["chol"].copy())
    s.run(40, adapt=False)
    after = s.state()
    assert after["scale"] == before[0] and np.array_equal(after["chol"], before[1])
    # the last exchange reports every walker's current state
    np.testing.assert_array_equal(after["all_params"][:, list(FREE)].shape, (4, 4))


def test_walkers_must_divide_over_ranks():
    with pytest.raises(hostlib.HostError):
        hostlib.HostSampler(3, FREE, STEP, hostlib.Exchange.callback(lambda r: np.concatenate([r, r]), 0, 2), run_block=lambda *a: None,
                            evaluate=lambda p: np.zeros(len(p)))


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gather = mcmc.torch_all_gather()
        ex = hostlib.Exchange.callback(lambda rows: gather(rows[None, :]).ravel(), rank, world)
        s, samples, lps = _run(rank, world, ex)
        st = s.state()
        np.savez(out_path, samples=samples, lps=lps, chol=st["chol"], scale=st["scale"], all_params=st["all_params"], all_logpost=st["all_logpost"])
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_one_rank(tmp_path):
    """world_size 2 over gloo: each rank advances half of the walkers; chains, proposal factor, step scale and the
    replicated ensemble state are the bits of the one-rank run."""
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    paths = [str(tmp_path / f"r{r}.npz") for r in range(2)]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, paths[r])) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    one, samples1, lps1 = _run(0, 1, hostlib.Exchange.local())
    st1 = one.state()
    r0, r1 = np.load(paths[0]), np.load(paths[1])
    np.testing.assert_array_equal(np.concatenate([r0["samples"], r1["samples"]], axis=1), samples1)
    np.testing.assert_array_equal(np.concatenate([r0["lps"], r1["lps"]], axis=1), lps1)
    for r in (r0, r1):                                       # replicated state agrees everywhere
        np.testing.assert_array_equal(r["chol"], st1["chol"])
        assert float(r["scale"]) == st1["scale"]
        np.testing.assert_array_equal(r["all_params"], st1["all_params"])
        np.testing.assert_array_equal(r["all_logpost"], st1["all_logpost"])
