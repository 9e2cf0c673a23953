"""CPU tests of the walker-parallel sampler (base_amd/mcmc.py): the counter-based RNG, the
adaptive block logic, and -- with world_size 2 over gloo -- that sharding the walkers over ranks
leaves every chain bit-identical (the only collective is the per-block all-gather)."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle
from base_amd import abi, mcmc, synth
from conftest import build_problem


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = mcmc.philox4x32(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_draws_are_standard_and_reproducible():
    z, u = mcmc.draws(7, 3, np.arange(20000), 4)
    z2, u2 = mcmc.draws(7, 3, np.arange(20000), 4)
    np.testing.assert_array_equal(z, z2)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and 0 < u.min() and u.max() < 1
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.03
    # a walker's stream does not depend on which other walkers are drawn with it
    z3, u3 = mcmc.draws(7, 3, np.array([5, 17]), 4)
    np.testing.assert_array_equal(z3, z[[5, 17]])
    np.testing.assert_array_equal(u3, u[[5, 17]])


def _problem():
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=150, seed=11)
    return pack_d, cl, oracle.Oracle(pack, stars, priors, options)


def _run(rank, world, gather, n_steps=60, block=20):
    pack_d, cl, orc = _problem()
    start = synth.walker_params(cl["truth"], 4, seed=42, scale=0.2)
    s = mcmc.WalkerSampler(start, mcmc.HostBlockRunner(orc.logpost), rank, world, gather, seed=99, block=block)
    s.initialise(orc.logpost)
    rec = []
    s.run(n_steps, rec)
    return s, rec


def test_sampler_moves_and_adapts():
    s, rec = _run(0, 1, None, n_steps=200, block=25)
    assert 0.05 < s.accepted / (200 * 4) < 0.95
    assert not np.allclose(s.chol, np.diag(np.diag(s.chol)))          # adapted: no longer diagonal
    assert np.all(np.isfinite(s.all_logpost))
    lps = np.concatenate([r[1] for r in rec])
    assert lps[-50:].mean() >= lps[:50].mean() - 1.0                   # not drifting away from the mode


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s, rec = _run(rank, world, mcmc.torch_all_gather())
        np.savez(out_path, samples=np.concatenate([r[0] for r in rec]), lps=np.concatenate([r[1] for r in rec]),
                 chol=s.chol, all_params=s.all_params, all_logpost=s.all_logpost)
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_one_rank(tmp_path):
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
        p.join(180)
        assert p.exitcode == 0
    one, rec = _run(0, 1, None)
    samples1 = np.concatenate([r[0] for r in rec])          # [steps, 4 walkers, d]
    lps1 = np.concatenate([r[1] for r in rec])
    r0, r1 = np.load(paths[0]), np.load(paths[1])
    np.testing.assert_array_equal(np.concatenate([r0["samples"], r1["samples"]], axis=1), samples1)
    np.testing.assert_array_equal(np.concatenate([r0["lps"], r1["lps"]], axis=1), lps1)
    for r in (r0, r1):                                       # replicated state agrees everywhere
        np.testing.assert_array_equal(r["chol"], one.chol)
        np.testing.assert_array_equal(r["all_params"], one.all_params)
        np.testing.assert_array_equal(r["all_logpost"], one.all_logpost)


def test_step_scale_stays_finite_when_every_step_moves():
    """A flat (or improper) posterior accepts everything: the global step scale must saturate, not overflow."""
    class AlwaysMoves:
        def run(self, params, logpost, ids, free, chol, seed, step0, n):
            W, d = params.shape[0], len(free)
            samples = params[None, :, free] + np.random.default_rng(step0).normal(size=(n, W, d)) * 1e-4
            return params, logpost, samples, np.zeros((n, W)), n * W
    start = np.tile(np.arange(abi.B9_NPARAM, dtype=float), (4, 1))
    s = mcmc.WalkerSampler(start, AlwaysMoves(), block=20)
    s.initialise(lambda p: np.zeros(len(p)))
    s.run(20 * 1500)
    assert s.scale == mcmc.SCALE_MAX and np.all(np.isfinite(s.chol))


def test_pipelined_run_equals_block_by_block():
    """WalkerSampler.run overlaps a block's host work with the next block's execution; the data dependencies are
    those of the unpipelined run_block loop, so states, proposal factor and scale are the same bits."""
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 4, n_stars=60, seed=2)
    orc = oracle.Oracle(pack, stars, priors, options)
    start = synth.walker_params(cl["truth"], 4, seed=3, scale=0.1)
    out = []
    for pipelined in (True, False):
        s = mcmc.WalkerSampler(start, mcmc.HostBlockRunner(orc.logpost), block=15, seed=11)
        s.initialise(orc.logpost)
        rec = []
        if pipelined:
            s.run(100, rec)          # 6 blocks of 15 + one of 10
            s.run(20, rec)
        else:
            for n in (15,) * 6 + (10,):
                rec.append(s.run_block(n))
            s.flush()
            for n in (15, 5):
                rec.append(s.run_block(n))
            s.flush()
        out.append((s.params.copy(), s.logpost.copy(), s.chol.copy(), s.scale, s.accepted, np.concatenate([r[0] for r in rec])))
    for a, b in zip(*out):
        assert np.array_equal(a, b)
