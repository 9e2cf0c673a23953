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


@pytest.mark.parametrize("env", [
    # the one-step fused launch (k_mcmc_step; B9_TREE_DEPTH=1 pins it: this small shape would otherwise take the tree launch)
    {"B9_TREE_DEPTH": "1", "B9_TILES_PER_BLOCK": "1", "B9_DERIVE_PARTS": "1"},                          # many workgroups, several occupancy rounds' worth of prologues
    {"B9_TREE_DEPTH": "1", "B9_TILES_PER_BLOCK": "2", "B9_DERIVE_PARTS": "3", "B9_DERIVE_ORDER": "-1"},  # derivation workgroups trail the grid
    {"B9_TREE_DEPTH": "1", "B9_TILES_PER_BLOCK": "5", "B9_CONTIGUOUS_TILES": "1"},                      # consecutive tiles, ragged last group
    {"B9_TREE_DEPTH": "1", "B9_TILES_PER_BLOCK": "7"},                                                  # strided tiles, groups without a last tile
    {"B9_TWO_LAUNCH_STEPS": "1"},                                                 # the two-launch step (what marginalised mode runs)
    {"B9_TREE_DEPTH": "1", "B9_DERIVE_ORDER": "0", "B9_HEAVY_PARTS": "3"},                               # heavy-star workgroups lead the grid; few, long heavy lists
    {"B9_TREE_DEPTH": "1", "B9_DERIVE_ORDER": "1", "B9_HEAVY_PARTS": "16", "B9_DERIVE_PARTS": "2"},      # (default order) many heavy parts, most of them idle
    # the tree-speculative launch (k_mcmc_tree): 2 / 3 steps per launch, blocks that are not a multiple of the depth
    {"B9_TREE_DEPTH": "2"},
    {"B9_TREE_DEPTH": "3"},                                                       # (more workgroups than one occupancy round at 5 walkers: still the same chain)
    {"B9_TREE_DEPTH": "3", "B9_TILES_PER_BLOCK": "5", "B9_DERIVE_PARTS": "1", "B9_HEAVY_PARTS": "2"},
    {"B9_TREE_DEPTH": "2", "B9_TILES_PER_BLOCK": "2", "B9_CONTIGUOUS_TILES": "1", "B9_DERIVE_PARTS": "3"},
    {},                                                                           # the automatic plan
])
@pytest.mark.parametrize("n_steps", [1, 2, 9, 10])
def test_fused_step_plans_all_give_the_same_chain(monkeypatch, env, n_steps):
    """The launch plan of the fused sampler step (tiles per workgroup, derivation parts, their place in the grid,
    strided or consecutive tiles) and the block length (1 step: no K(t) ever takes a decision; 2: one does)
    only regroup the same work: every plan reproduces the host twin's chain."""
    from base_amd import engine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=3000, wd_frac=0.04, small=False, seed=77)
    eng = engine.Engine(pack, stars, priors, options)                              # 3000 stars -> 12 tiles
    free = np.array(mcmc.DEFAULT_FREE)
    start = synth.walker_params(cl["truth"], 5, seed=8, scale=0.1)
    start[4, abi.P_LOGAGE] = pack_d["log_age"][-1] - 1e-4                          # proposals of this walker leave the grid now and then
    ids = np.array([3, 1, 4, 1, 5])                                                # two walkers share an RNG stream id: still independent state
    chol = np.diag([3e-4, 2e-3, 8e-4, 6e-4])
    lp0 = eng.logpost(start)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, ids, free, chol, 5, 123456789012, n_steps)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, ids, free, chol, 5, 123456789012, n_steps)
    assert dev[4] == host[4]
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)
    np.testing.assert_allclose(dev[0], host[0], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[1], host[1], rtol=1e-10)
    # a second block continues the first (state and RNG counters carry over): 2 x n == one block of 2 n
    two = mcmc.DeviceBlockRunner(eng).run(dev[0], dev[1], ids, free, chol, 5, 123456789012 + n_steps, n_steps)
    one = mcmc.DeviceBlockRunner(eng).run(start, lp0, ids, free, chol, 5, 123456789012, 2 * n_steps)
    np.testing.assert_allclose(two[0], one[0], rtol=1e-12, atol=1e-13)
    assert dev[4] + two[4] == one[4]


@pytest.mark.parametrize("depth", ["1", "2", "3"])
@pytest.mark.parametrize("n_pops,n_y", [(1, 1), (2, 3)])
def test_tree_candidates_in_other_grid_cells(monkeypatch, depth, n_pops, n_y):
    """The derivation role -- of the tree launch (depth 2, 3) and of the one-step fused launch (depth 1:
    step_derive_ahead) -- reads its tables AHEAD for the grid cell of the previous state; walkers that sit on cell borders
    (age, FeH, Y) and take steps of a cell's size put many candidates in another cell, which then repeat the reads for
    their own: the chain is the host twin's either way."""
    from base_amd import engine
    monkeypatch.setenv("B9_TREE_DEPTH", depth)
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 5, n_stars=1500, wd_frac=0.02, n_y=n_y, n_pops=n_pops, small=False, seed=5)
    eng = engine.Engine(pack, stars, priors, options)
    free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS] + ([abi.P_Y, abi.P_Y2, abi.P_LAMBDA] if n_pops == 2 else []))
    W = 4
    start = synth.walker_params(cl["truth"], W, seed=3, scale=0.05, n_pops=n_pops)
    ages, fehs = np.asarray(pack_d["log_age"]), np.asarray(pack_d["feh"])
    for w in range(W):                                                             # on (or a hair beside) a grid line in age and FeH
        start[w, abi.P_LOGAGE] = ages[len(ages) // 2 + w] + (w - 1) * 1e-9
        start[w, abi.P_FEH] = fehs[1 + w % (len(fehs) - 2)] + (1 - w) * 1e-9
    d_age, d_feh = float(np.min(np.diff(ages))), float(np.min(np.diff(fehs)))
    steps = [0.7 * d_age, 0.5 * d_feh, 5e-4, 5e-4] + ([0.05, 0.05, 2e-3] if n_pops == 2 else [])
    chol = np.diag(steps)
    lp0 = eng.logpost(start)
    assert eng.step_depth(W) == int(depth)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, np.arange(W), free, chol, 21, 0, 40)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, np.arange(W), free, chol, 21, 0, 40)
    assert dev[4] == host[4]
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)
    cells = {(int(np.searchsorted(ages, r[0], side="right")), int(np.searchsorted(fehs, r[1], side="right"))) for r in dev[2].reshape(-1, len(free))}
    assert len(cells) >= 3, "the chain should have visited several grid cells"


def test_fused_step_many_walkers_and_two_populations_several_rounds(monkeypatch):
    """More walkers than one occupancy round holds at 1 tile per workgroup: later workgroups may take the
    published decision instead of re-deriving it -- same chain."""
    from base_amd import engine
    monkeypatch.setenv("B9_TILES_PER_BLOCK", "1")
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 5, n_stars=6000, wd_frac=0.02, n_y=3, n_pops=2, small=False, seed=2)
    eng = engine.Engine(pack, stars, priors, options)
    free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS, abi.P_Y, abi.P_Y2, abi.P_LAMBDA])
    W = 40                                                                         # 24 tiles x 40 walkers = 960 hot workgroups > 512 slots
    start = synth.walker_params(cl["truth"], W, seed=3, scale=0.1, n_pops=2)
    chol = np.diag([2e-4, 1e-3, 5e-4, 5e-4, 3e-4, 3e-4, 2e-3])
    lp0 = eng.logpost(start)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, np.arange(W), free, chol, 11, 0, 6)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, np.arange(W), free, chol, 11, 0, 6)
    assert dev[4] == host[4] and dev[4] > 0
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)


@pytest.mark.parametrize("n_pops,n_y", [(1, 1), (2, 3)])
def test_tree_and_one_step_launches_give_the_same_bits(monkeypatch, n_pops, n_y):
    """With the star-to-partial-sum grouping pinned (tiles per workgroup), the tree launch at depth 2 and 3 walks exactly the
    one-step launch's sums (per node the same words in the same lanes, the same cross-lane tree -- packed seven at a time in
    the tree walk) and takes exactly its decisions: the chains are equal bit for bit, not just to the tolerance."""
    from base_amd import engine
    monkeypatch.setenv("B9_TILES_PER_BLOCK", "2")
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=5000, wd_frac=0.04, n_y=n_y, n_pops=n_pops, small=False, seed=21)
    free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS] + ([abi.P_Y, abi.P_Y2, abi.P_LAMBDA] if n_pops == 2 else []))
    W = 2
    start = synth.walker_params(cl["truth"], W, seed=9, scale=0.05, n_pops=n_pops)
    chol = np.diag([3e-4, 2e-3, 8e-4, 6e-4] + ([3e-4, 3e-4, 2e-3] if n_pops == 2 else []))
    runs = {}
    for depth in ("1", "2", "3"):
        monkeypatch.setenv("B9_TREE_DEPTH", depth)
        eng = engine.Engine(pack, stars, priors, options)
        assert eng.step_depth(W) == int(depth)
        lp0 = eng.logpost(start)
        runs[depth] = eng.mcmc_run_block(start, lp0, np.arange(W), free, chol, 17, 3, 31)
    assert runs["1"][4] > 0
    for depth in ("2", "3"):
        for a, b in zip(runs["1"][:4], runs[depth][:4]):
            np.testing.assert_array_equal(a, b)
        assert runs["1"][4] == runs[depth][4]


def test_tree_depth_follows_the_catalogue_size():
    """Speculation pays only while the chip is under-filled: one walker on a small catalogue takes three steps per launch, on a
    catalogue whose tiles saturate the CUs the one-step launch (make_tree_plan's cost estimate).  A pinned depth runs wherever
    the catalogue's canonical tile groups are few enough for one walk to read (<= 48 per node); the grouping fixes how a
    log-posterior rounds and is never changed for the tree's sake, so on the large catalogue the pin has no effect."""
    from base_amd import engine
    pack_d = synth.make_pack("parsec", 8)
    truth = synth.default_params(pack_d)
    depth = {}
    for n_stars in (10000, 400000):
        cl = synth.make_cluster(pack_d, n_stars, seed=3, truth=truth)
        eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
        depth[n_stars] = eng.step_depth(1)
        eng.set_tuning(tree_depth=2)
        assert eng.step_depth(1) == (2 if n_stars == 10000 else 1)
        eng.close()
    assert depth == {10000: 3, 400000: 1}


def test_device_chain_is_bit_reproducible():
    """No atomics on the data path, fixed summation orders, counter-based RNG: the same block run twice is the
    same bits (also when several occupancy rounds let late workgroups take the published decision)."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=20000, wd_frac=0.03, small=False, seed=9)
    eng = engine.Engine(pack, stars, priors, options)
    free, chol = np.array(mcmc.DEFAULT_FREE), np.diag([5e-5, 3e-4, 1e-4, 1e-4])
    start = synth.walker_params(cl["truth"], 16, seed=5, scale=0.05)
    lp0 = eng.logpost(start)
    a = eng.mcmc_run_block(start, lp0, np.arange(16), free, chol, 3, 0, 300)
    b = eng.mcmc_run_block(start, lp0, np.arange(16), free, chol, 3, 0, 300)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert 0 < a[4] < 300 * 16


@pytest.mark.parametrize("n_stars", [1, 2, 63, 64, 65, 255, 256, 257, 511, 513])
def test_fused_step_at_chunk_and_tile_boundaries(n_stars):
    """Clusters of exactly / just under / just over one 64-star chunk and one 256-star tile (binaries and singles
    are chunked separately, so these sizes exercise half-empty waves and all-empty tiles)."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 5, n_stars=n_stars, wd_frac=0.1 if n_stars > 10 else 0.0, small=False, seed=n_stars)
    eng = engine.Engine(pack, stars, priors, options)
    free, chol = np.array(mcmc.DEFAULT_FREE), np.diag([3e-3, 2e-2, 8e-3, 6e-3])
    start = synth.walker_params(cl["truth"], 3, seed=8, scale=0.1)
    lp0 = eng.logpost(start)
    np.testing.assert_allclose(lp0, oracle.Oracle(pack, stars, priors, options).logpost(start), rtol=1e-9)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, np.arange(3), free, chol, 5, 0, 12)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, np.arange(3), free, chol, 5, 0, 12)
    assert dev[4] == host[4]
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)


def test_pipelined_blocks_continue_on_the_device():
    """B9_BLOCK_CONTINUE | B9_BLOCK_ASYNC: blocks enqueued back to back from the device-resident state give the
    chain of the synchronous block-by-block calls (and of one long block), whatever the block lengths."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=2000, wd_frac=0.03, small=False, seed=13)
    eng = engine.Engine(pack, stars, priors, options)
    free, chol = np.array(mcmc.DEFAULT_FREE), np.diag([3e-4, 2e-3, 8e-4, 6e-4])
    start = synth.walker_params(cl["truth"], 6, seed=2, scale=0.1)
    lp0 = eng.logpost(start)
    sizes = [7, 1, 12, 2, 9]
    # synchronous, block by block, host state handed over
    p, lp, acc, samp = start, lp0, 0, []
    s0 = 1000
    for n in sizes:
        p, lp, s, l, a = eng.mcmc_run_block(p, lp, np.arange(6), free, chol * (1 + 0.1 * len(samp)), 4, s0, n)
        samp.append(s); acc += a; s0 += n
    # pipelined: two outstanding, continuing on the device
    hs, s0, got, acc2 = [], 1000, [], 0
    for k, n in enumerate(sizes):
        hs.append(eng.mcmc_submit(start, lp0, np.arange(6), free, chol * (1 + 0.1 * k), 4, s0, n, cont=k > 0))
        s0 += n
        if len(hs) == 2:
            r = eng.mcmc_collect(hs.pop(0)); got.append(r[2]); acc2 += r[4]
    while hs:
        r = eng.mcmc_collect(hs.pop(0)); got.append(r[2]); acc2 += r[4]
    assert acc2 == acc
    for a, b in zip(samp, got):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(r[0], p)
    np.testing.assert_array_equal(r[1], lp)
    # misuse is reported, not UB
    h1 = eng.mcmc_submit(start, lp0, np.arange(6), free, chol, 4, 0, 3)
    h2 = eng.mcmc_submit(start, lp0, np.arange(6), free, chol, 4, 3, 3, cont=True)
    with pytest.raises(RuntimeError):
        eng.mcmc_submit(start, lp0, np.arange(6), free, chol, 4, 6, 3, cont=True)        # a third outstanding block
    with pytest.raises(RuntimeError):
        eng.mcmc_collect(h2)                                                              # out of order
    eng.mcmc_collect(h1); eng.mcmc_collect(h2)
    with pytest.raises(RuntimeError):
        eng.mcmc_submit(start[:3], lp0[:3], np.arange(3), free, chol, 4, 0, 3, cont=True)  # other walker count


def test_sampler_device_pipeline_equals_unpipelined(monkeypatch):
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=1500, small=False, seed=3)
    eng = engine.Engine(pack, stars, priors, options)
    start = synth.walker_params(cl["truth"], 8, seed=1, scale=0.05)
    out = []
    for pipe in (True, False):
        if not pipe:
            monkeypatch.setenv("B9_NO_BLOCK_PIPELINE", "1")
        s = mcmc.WalkerSampler(start, mcmc.DeviceBlockRunner(eng), block=25, seed=5)
        s.initialise(eng.logpost)
        rec = []
        s.run(260, rec)
        s.run(40, rec)
        out.append((s.params.copy(), s.logpost.copy(), s.chol.copy(), s.scale, s.accepted, s.step, np.concatenate([r[0] for r in rec])))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def test_an_outstanding_block_owns_the_work_buffers():
    """While an asynchronous block is enqueued, calls that would overwrite or re-allocate the context's work buffers
    (b9_logpost evaluates into the candidate isochrones the block is reading; a block with another walker count resizes
    them) are refused with B9_ERR_STATE instead of corrupting the chain; after the block is collected they work again."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=3000, wd_frac=0.02, small=False, seed=3)
    eng = engine.Engine(pack, stars, priors, options)
    free = np.array(mcmc.DEFAULT_FREE)
    start = synth.walker_params(cl["truth"], 4, seed=1, scale=0.05)
    lp0 = eng.logpost(start)
    chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free])
    ref = eng.mcmc_run_block(start, lp0, np.arange(4), free, chol, 3, 0, 40)
    h = eng.mcmc_submit(start, lp0, np.arange(4), free, chol, 3, 0, 40, record=True, asynchronous=True)
    with pytest.raises(engine.B9Error) as e1:
        eng.logpost(start)
    assert e1.value.code == abi.B9_ERR_STATE
    with pytest.raises(engine.B9Error):
        eng.mcmc_submit(start[:2], lp0[:2], np.arange(2), free, chol, 3, 0, 5, record=False, asynchronous=True)
    with pytest.raises(engine.B9Error):
        eng.set_options(abi.make_options(abi.MODE_MARGINALISED, 1, 2, 2))
    got = eng.mcmc_collect(h)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    np.testing.assert_array_equal(eng.logpost(start), lp0)
    eng.close()
