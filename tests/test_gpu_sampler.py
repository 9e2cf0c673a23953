"""GPU tests of the C++ walker-parallel driver on its product path: device-resident pipelined blocks, summary rows
condensed by the block's last launch, RCCL exchange.  (Its host logic alone is covered on CPU by test_sampler_host.py.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from base_amd import abi, hostlib, mcmc, synth
from conftest import build_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FREE = (abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS)
STEP = [mcmc.DEFAULT_STEP[k] for k in FREE]


def _gpu_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("n_steps,d", [(1, 4), (37, 4), (128, 2), (300, 5)])
def test_device_summary_rows_equal_the_host_statement(n_steps, d):
    """b9_mcmc_block::rows (k_mcmc_finish) against b9h::summary_rows of the downloaded chain: the same bits."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=700, wd_frac=0.03, small=False, seed=5)
    eng = engine.Engine(pack, stars, priors, options)
    W = 6
    free = np.array((abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS, abi.P_CARBONICITY)[:d], dtype=np.int32)
    start = synth.walker_params(cl["truth"], W, seed=3, scale=0.05)
    lp = eng.logpost(start)
    chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 2.0
    origin = start[:, free].mean(axis=0)
    h = eng.mcmc_submit(start, lp, np.arange(W), free, chol, 21, 0, n_steps, record=True, asynchronous=False, row_origin=origin)
    params, logpost, samples, lps, n_acc = eng.mcmc_collect(h)
    want = hostlib.summary_rows(samples, params, logpost, origin)
    np.testing.assert_array_equal(h["rows"], want)
    assert h["rows"][:, 14].tolist() == [n_steps] * W
    # rows without a host chain record: the samples exist on the device only
    h2 = eng.mcmc_submit(start, lp, np.arange(W), free, chol, 21, 0, n_steps, record=False, asynchronous=False, row_origin=origin)
    eng.mcmc_collect(h2)
    np.testing.assert_array_equal(h2["rows"], want)


def _device_run(eng, exchange, start, n_walkers=8, burn=150, main=50, block=50, seed=77):
    s = hostlib.HostSampler(n_walkers, FREE, STEP, exchange, seed=seed, block=block, engine=eng)
    s.initialise(start)
    a = s.run(burn, adapt=True, record=True)
    b = s.run(main, adapt=False, record=True)
    return s.state(), np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]])


def test_device_sampler_equals_the_host_twin_and_one_rank_rccl():
    """The C++ sampler on the GPU (fused steps, pipelined blocks, device rows) produces the chain of the same sampler
    driven block by block through the numpy twin of the device step (HostBlockRunner over the engine's log-posterior):
    same proposals, same decisions, same adaptation (to the ulp of the two normal generators).  A one-rank RCCL exchange (ncclAllGather reading the rows in
    HBM) changes nothing."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=3000, wd_frac=0.02, small=False, seed=8)
    eng = engine.Engine(pack, stars, priors, options)
    start = synth.walker_params(cl["truth"], 8, seed=42, scale=0.05)
    st_dev, samples_dev, lps_dev = _device_run(eng, hostlib.Exchange.local(), start)
    assert 0.03 < st_dev["accepted_local"] / (200 * 8) < 0.97
    twin = mcmc.HostBlockRunner(eng.logpost)
    s = hostlib.HostSampler(8, FREE, STEP, hostlib.Exchange.local(), seed=77, block=50, run_block=twin.run, evaluate=eng.logpost)
    s.initialise(start)
    a = s.run(150, adapt=True, record=True)
    b = s.run(50, adapt=False, record=True)
    # (the twin's normals come from numpy's log / sin / cos, the device's from its own: equal to an ulp, as in test_gpu_mcmc)
    np.testing.assert_allclose(np.concatenate([a[0], b[0]]), samples_dev, rtol=1e-12, atol=1e-13)
    st_twin = s.state()
    np.testing.assert_allclose(st_twin["chol"], st_dev["chol"], rtol=1e-8, atol=1e-14)
    assert abs(st_twin["scale"] / st_dev["scale"] - 1.0) < 1e-8
    # the chain's log-posteriors agree with the oracle at the visited points (spot check)
    orc = oracle.Oracle(pack, stars, priors, options)
    rows = np.repeat(start[:1], 5, axis=0)
    rows[:, list(FREE)] = samples_dev[[10, 60, 110, 160, 199], 0]
    want = orc.logpost(rows)
    got = lps_dev[[10, 60, 110, 160, 199], 0]
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= 1e-9
    # one-rank RCCL group
    ex = hostlib.Exchange.rccl(0, 1, eng.device_id(), directory=None)
    assert "RCCL" in ex.name
    st_rccl, samples_rccl, _ = _device_run(eng, ex, start)
    np.testing.assert_array_equal(samples_rccl, samples_dev)
    np.testing.assert_array_equal(st_rccl["chol"], st_dev["chol"])


@pytest.mark.parametrize("mode", ["given_mass", "marginalised"])
def test_two_ranks_on_one_gpu_match_one_rank(mode):
    """World size 2 through the PRODUCT block runner (device-resident fused steps, pipelined blocks) on one GPU: two
    threads, each with its own context and half of the walkers; only the all-gather is bridged by the test (a callback
    exchange -- RCCL refuses two ranks on one device).  Chains, proposal factor, scale and the replicated ensemble
    state are the bits of the one-rank run -- in the marginalised mode too, where this catalogue (47 star chunks) splits
    every chunk's window over 8 workgroups: the split is a function of the catalogue, not of the walkers on the GPU."""
    import threading
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=3000, wd_frac=0.02, small=False, seed=8)
    if mode == "marginalised":
        options = abi.make_options(abi.MODE_MARGINALISED, 1, 2, 3)
    start = synth.walker_params(cl["truth"], 8, seed=42, scale=0.05)
    eng1 = engine.Engine(pack, stars, priors, options)
    st1, samples1, lps1 = _device_run(eng1, hostlib.Exchange.local(), start)
    eng1.close()

    slots, bar = [None, None], threading.Barrier(2, timeout=120)

    def make_gather(rank):
        def gather(rows):
            slots[rank] = np.array(rows, copy=True)
            bar.wait()
            out = np.concatenate([slots[0], slots[1]])
            bar.wait()                                   # nobody overwrites a slot the other rank still reads
            return out
        return gather

    results, errors = [None, None], []

    def rank_main(rank):
        try:
            eng = engine.Engine(pack, stars, priors, options)
            ex = hostlib.Exchange.callback(make_gather(rank), rank, 2)
            results[rank] = _device_run(eng, ex, start)
            eng.close()
        except BaseException as e:                       # noqa: BLE001  (reported by the main thread)
            errors.append(e)
            bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    (st_a, samples_a, lps_a), (st_b, samples_b, lps_b) = results
    np.testing.assert_array_equal(np.concatenate([samples_a, samples_b], axis=1), samples1)
    np.testing.assert_array_equal(np.concatenate([lps_a, lps_b], axis=1), lps1)
    for st in (st_a, st_b):
        np.testing.assert_array_equal(st["chol"], st1["chol"])
        assert st["scale"] == st1["scale"]
        np.testing.assert_array_equal(st["all_params"], st1["all_params"])
        np.testing.assert_array_equal(st["all_logpost"], st1["all_logpost"])


def _two_ranks_one_gpu(pack, stars, priors, options, start, tuning, **run_kw):
    """World size 2 through the product block runner on ONE GPU (two threads, a context each, half of the walkers; the
    all-gather bridged by a callback exchange).  Returns the two ranks' (state, samples, lps)."""
    import threading
    from base_amd import engine
    slots, bar = [None, None], threading.Barrier(2, timeout=300)

    def make_gather(rank):
        def gather(rows):
            slots[rank] = np.array(rows, copy=True)
            bar.wait()
            out = np.concatenate([slots[0], slots[1]])
            bar.wait()
            return out
        return gather

    results, errors = [None, None], []

    def rank_main(rank):
        try:
            eng = engine.Engine(pack, stars, priors, options)
            if tuning:
                eng.set_tuning(**tuning)
            ex = hostlib.Exchange.callback(make_gather(rank), rank, 2)
            results[rank] = _device_run(eng, ex, start, **run_kw)
            eng.close()
        except BaseException as e:                       # noqa: BLE001  (reported by the main thread)
            errors.append(e)
            bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    return results


@pytest.mark.parametrize("n_walkers", [8, 16])
def test_rank_count_invariance_at_50k_stars(n_walkers):
    """A walker's chain is the same BITS whatever the number of ranks its ensemble is spread over -- with nothing pinned.
    The star-to-partial-sum grouping (canonical tile groups: DESIGN.md section 3) is a function of the catalogue, the pack and
    the device only; a launch plan only chooses how many whole groups a workgroup takes.  50 000 stars: 8 walkers on one GPU
    against 4 + 4 on two ranks (same groups per workgroup, half the workgroups), and 16 walkers on one GPU -- two groups
    = 6 tiles per hot workgroup -- against 8 + 8 at one group = 3 tiles: the plans differ, the bits do not."""
    from base_amd import engine
    cfg = synth.make_baseline_config("C2")
    pack, stars, priors, options = cfg["pack"], cfg["stars"], cfg["priors"], cfg["options"]
    start = synth.walker_params(cfg["truth"], n_walkers, seed=42, scale=0.02)
    kw = dict(n_walkers=n_walkers, burn=60, main=20, block=20)
    eng1 = engine.Engine(pack, stars, priors, options)
    t_all, t_half = eng1.step_tiles_per_block(n_walkers), eng1.step_tiles_per_block(n_walkers // 2)
    if n_walkers == 16:
        assert t_all != t_half, "the two launch plans no longer differ: pick another shape for this case"
    st1, samples1, lps1 = _device_run(eng1, hostlib.Exchange.local(), start, **kw)
    eng1.close()
    (sa, xa, la), (sb, xb, lb) = _two_ranks_one_gpu(pack, stars, priors, options, start, None, **kw)
    np.testing.assert_array_equal(np.concatenate([xa, xb], axis=1), samples1)
    np.testing.assert_array_equal(np.concatenate([la, lb], axis=1), lps1)
    for st in (sa, sb):
        np.testing.assert_array_equal(st["chol"], st1["chol"])
        assert st["scale"] == st1["scale"]
        np.testing.assert_array_equal(st["all_logpost"], st1["all_logpost"])


def _oracle_delta(orc, template_row, free, samples, lps):
    """max relative |delta| between recorded chain log-posteriors and the oracle at the recorded positions; every
    DISTINCT visited state is evaluated once (a rejected step repeats its predecessor's row)."""
    flat = samples.reshape(-1, samples.shape[-1])
    uniq, inverse = np.unique(flat, axis=0, return_inverse=True)
    rows = np.repeat(np.asarray(template_row, dtype=np.float64)[None, :], len(uniq), axis=0)
    rows[:, list(free)] = uniq
    want = orc.logpost(rows)[inverse.ravel()]
    got = lps.reshape(-1)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    fin = np.isfinite(want)
    return float(np.max(np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin])))), len(uniq)


@pytest.mark.parametrize("name", ["C0", "C1", "C2", "C3", "C4"])
def test_sampler_logposts_match_oracle_full_size(name):
    """The TIMED path -- k_mcmc_step driven by the C++ sampler (b9h::WalkerSampler: fused one-launch steps, pipelined
    device-resident blocks) -- at the FULL size of every BASELINE.json configuration (one GPU's share): every
    log-posterior the chain records over 60 steps (30 adapting, 30 frozen; two blocks each) equals the CPU oracle's at
    the recorded position to 1e-9 relative.  This is the fused step's own reduction (per-wave partials of two parities,
    heavy-star partials of two candidates, first-wave decision, published-decision shortcut) at benchmark size."""
    from base_amd import engine
    cfg = synth.make_baseline_config(name)
    eng = engine.Engine(cfg["pack"], cfg["stars"], cfg["priors"], cfg["options"])
    W, free = cfg["walkers"], cfg["free"]
    start = synth.walker_params(cfg["truth"], W, seed=7, n_pops=cfg["n_pops"], scale=0.02)
    s = hostlib.HostSampler(W, free, [mcmc.DEFAULT_STEP[k] for k in free], hostlib.Exchange.local(), seed=11, block=15, engine=eng)
    s.initialise(start)
    a = s.run(30, adapt=True, record=True)
    b = s.run(30, adapt=False, record=True)
    samples, lps = np.concatenate([a[0], b[0]]), np.concatenate([a[1], b[1]])
    assert samples.shape == (60, W, len(free))
    orc = oracle.Oracle(cfg["pack"], cfg["stars"], cfg["priors"], cfg["options"])
    worst, n_states = 0.0, 0
    for w in range(W):                       # parameters that are not sampled stay at the walker's starting values
        err, n = _oracle_delta(orc, start[w], free, samples[:, w], lps[:, w])
        worst, n_states = max(worst, err), n_states + n
    assert n_states > W, "the chains never moved: nothing but the starting state was checked"
    assert worst <= 1e-9, (name, worst)
    # the ensemble state the sampler reports after the run (what bench.py checks after its timed region)
    st = s.state()
    want = orc.logpost(st["all_params"])
    assert np.max(np.abs(st["all_logpost"] - want) / np.maximum(1.0, np.abs(want))) <= 1e-9
    eng.close()


def test_marginalised_mode_through_the_sampler():
    """Marginalised mode runs the same pipelined blocks as given-mass mode (two launches per step instead of one):
    B9_BLOCK_ASYNC | CONTINUE, summary rows condensed on the device (k_chain_rows) and read from HBM by a one-rank RCCL
    all-gather.  Chain log-posteriors equal the oracle's marginalised ones; the device rows equal the host statement;
    two continued blocks equal one block of twice the length; the RCCL exchange changes nothing."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, _ = build_problem("dsed", 8, n_stars=200, seed=4)
    options = abi.make_options(abi.MODE_MARGINALISED, 1, 2, 2)
    eng = engine.Engine(pack, stars, priors, options)
    start = synth.walker_params(cl["truth"], 4, seed=1, scale=0.05)
    s = hostlib.HostSampler(4, FREE, STEP, hostlib.Exchange.local(), seed=5, block=10, engine=eng)
    s.initialise(start)
    samples, lps = s.run(30, adapt=True, record=True)
    orc = oracle.Oracle(pack, stars, priors, options)
    rows = np.repeat(start[:1], 3, axis=0)
    rows[:, list(FREE)] = samples[[3, 17, 29], 1]
    want = orc.logpost(rows)
    assert np.max(np.abs(lps[[3, 17, 29], 1] - want) / np.maximum(1.0, np.abs(want))) <= 1e-9
    assert s.state()["steps"] == 30
    # one-rank RCCL exchange reading the rows in HBM: the same chain
    ex = hostlib.Exchange.rccl(0, 1, eng.device_id(), directory=None)
    s2 = hostlib.HostSampler(4, FREE, STEP, ex, seed=5, block=10, engine=eng)
    s2.initialise(start)
    samples2, lps2 = s2.run(30, adapt=True, record=True)
    np.testing.assert_array_equal(samples2, samples)
    np.testing.assert_array_equal(lps2, lps)
    # block level: device rows == host statement; async + continue == one long block
    free = np.array(FREE, dtype=np.int32)
    chol = np.diag(STEP) * 2.0
    lp0 = eng.logpost(start)
    origin = start[:, free].mean(axis=0)
    one = eng.mcmc_submit(start, lp0, np.arange(4), free, chol, 9, 0, 16, record=True, asynchronous=False, row_origin=origin)
    p1, l1, x1, y1, a1 = eng.mcmc_collect(one)
    np.testing.assert_array_equal(one["rows"], hostlib.summary_rows(x1, p1, l1, origin))
    ha = eng.mcmc_submit(start, lp0, np.arange(4), free, chol, 9, 0, 8, record=True, asynchronous=True, row_origin=origin)
    hb = eng.mcmc_submit(start, lp0, np.arange(4), free, chol, 9, 8, 8, record=True, cont=True, asynchronous=True, row_origin=origin)
    pa, la, xa, ya, aa = eng.mcmc_collect(ha)
    pb, lb, xb, yb, ab = eng.mcmc_collect(hb)
    np.testing.assert_array_equal(np.concatenate([xa, xb]), x1)
    np.testing.assert_array_equal(np.concatenate([ya, yb]), y1)
    np.testing.assert_array_equal(pb, p1)
    assert aa + ab == a1
    np.testing.assert_array_equal(hb["rows"], hostlib.summary_rows(xb, pb, lb, origin))
    # a marginalised block cannot continue a given-mass block
    eng.set_options(abi.make_options(abi.MODE_GIVEN_MASS, 1, 2, 2))
    hg = eng.mcmc_submit(start, eng.logpost(start), np.arange(4), free, chol, 9, 0, 4, record=False, asynchronous=False)
    eng.mcmc_collect(hg)
    eng.set_options(options)
    with pytest.raises(engine.B9Error):
        eng.mcmc_submit(start, lp0, np.arange(4), free, chol, 9, 4, 4, record=False, cont=True, asynchronous=False)
    eng.close()


def test_bench_self_launch_two_ranks_on_one_box():
    """`python bench.py --gpus 2` invoked plainly starts its own ranks and prints rank 0's line (needs 2 GPUs)."""
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and "RCCL" in line["config"]["collective"]
    assert line["config"]["rccl_ranks"] == 2 and len(set(line["config"]["devices"])) == 2
    assert line["config"]["walkers_total"] == 16 and line["value"] > 0


def test_cli_two_gpus_chains_equal_one_gpu(tmp_path):
    """singlePopMcmc --gpus 2 (one process per GPU, RCCL all-gather of the rows in HBM): the merged .res is the
    one-GPU run's, row for row (needs 2 GPUs)."""
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs")
    from base_amd import host_build
    host_build.build_host()
    pack_d, cl, *_ = build_problem("parsec", 8, n_stars=2000, small=False, seed=12)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    phot = synth.write_phot(cl, pack_d["filters"], str(tmp_path / "c.phot"))
    outs = []
    for gpus in (1, 2):
        base = str(tmp_path / f"run{gpus}")
        yml = synth.write_yaml(str(tmp_path / f"b{gpus}.yaml"), phot, root, base, cl["truth"])
        exe = os.path.join(ROOT, "base_amd", "host", "bin", "singlePopMcmc")
        r = subprocess.run([exe, "--config", yml, "--walkers", "4", "--gpus", str(gpus), "--burnIter", "120", "--runIter", "60",
                            "--block", "30", "--seed", "5"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(open(base + ".res").read())
    assert outs[0] == outs[1] and outs[0].count("\n") == 1 + 180 * 4


def test_cli_forced_ranks_on_one_gpu_gives_the_same_res(tmp_path):
    """The multi-rank route rehearsed on ONE GPU: `singlePopMcmc --gpus 1 --forceRanks` starts a child rank before any GPU
    call, bootstraps RCCL through the id file (ncclCommInitRank, world 1), all-gathers the device rows through it,
    writes .res.part0 and merges it -- and the .res is byte for byte the plain one-process run's."""
    from base_amd import host_build
    host_build.build_host()
    pack_d, cl, *_ = build_problem("parsec", 8, n_stars=2000, small=False, seed=12)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    phot = synth.write_phot(cl, pack_d["filters"], str(tmp_path / "c.phot"))
    exe = os.path.join(ROOT, "base_amd", "host", "bin", "singlePopMcmc")
    outs, errs = [], []
    for tag, extra in (("plain", []), ("forced", ["--gpus", "1", "--forceRanks"])):
        base = str(tmp_path / f"run_{tag}")
        yml = synth.write_yaml(str(tmp_path / f"b_{tag}.yaml"), phot, root, base, cl["truth"])
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "B9_RANK")}
        r = subprocess.run([exe, "--config", yml, "--walkers", "4", "--burnIter", "120", "--runIter", "60", "--block", "30", "--seed", "5"] + extra,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(open(base + ".res").read())
        errs.append(r.stderr)
        assert not os.path.exists(base + ".res.part0")
    assert outs[0] == outs[1] and outs[0].count("\n") == 1 + 180 * 4
    assert outs[0].split()[0] == "logAge" and open(base + ".res.meta").read().startswith("base9_hip ABI ")
    assert "RCCL" in errs[1] and "communicator of 1 rank(s)" in errs[1] and "RCCL" not in errs[0]


def test_bench_forced_ranks_on_one_gpu():
    """`bench.py --gpus 1 --force-ranks`: self-launched rank, RCCL communicator of one rank, device rows gathered by
    ncclAllGather; the line names the collective and carries what the communicator itself reports."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "B9_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-ranks", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-marginalised"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["ranks"] == 1 and cfg["forced_ranks"] is True
    assert "RCCL" in cfg["collective"] and cfg["rccl_ranks"] == 1 and len(cfg["devices"]) == 1 and ":" in cfg["devices"][0]
    assert line["value"] > 0 and line["roofline"]["useful_frac"] > 0


def test_bench_under_torchrun_with_an_rccl_communicator():
    """The driver's own launch form -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- with N = 1 and
    the multi-rank route forced (B9_FORCE_RANKS): the rank takes RANK / WORLD_SIZE / LOCAL_RANK from torchrun, names the
    RCCL id file after torchrun's run id (no B9_LAUNCH_NONCE here), brings a communicator up and gathers device rows."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "B9_RANK", "B9_LAUNCH_NONCE", "B9_DIST_DIR")}
    env["B9_FORCE_RANKS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-marginalised", "--no-sustained"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and cfg["ranks"] == 1 and "RCCL" in cfg["collective"] and cfg["rccl_ranks"] == 1 and len(cfg["devices"]) == 1
    assert line["value"] > 0 and line["sustained"] is None
