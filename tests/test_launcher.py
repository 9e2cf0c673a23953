"""CPU tests of the two self-launchers (singlePopMcmc --gpus N in C++, bench.py --gpus N in Python): start-up deadline,
exit code of a failed launch, clean-up -- and of bench.py's guard against stale profile counters.  No GPU is needed: the
test hook B9_TEST_STALL parks a rank before it touches one."""
import glob
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "base_amd", "host", "bin", "singlePopMcmc")


@pytest.fixture(scope="module", autouse=True)
def _built():
    from base_amd import host_build
    host_build.build_host()


def _dist_dirs():
    return set(glob.glob("/tmp/b9dist_*")) | set(glob.glob(os.path.join(os.environ.get("TMPDIR", "/tmp"), "b9dist_*")))


def _run(cmd, env_extra, timeout=120):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "B9_RANK"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    return r, time.monotonic() - t0


@pytest.mark.parametrize("which", ["cli", "bench"])
def test_stalled_rank_ends_the_launch_at_the_deadline(which):
    """A rank that never brings its communicator up (here: parked before anything) makes the launcher give up after
    B9_LAUNCH_TIMEOUT_S, kill it and exit 124 -- instead of waiting for ever, as ncclCommInitRank would."""
    before = _dist_dirs()
    cmd = [EXE, "--gpus", "1", "--forceRanks"] if which == "cli" else [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-ranks"]
    r, dt = _run(cmd, {"B9_TEST_STALL": "start:0", "B9_LAUNCH_TIMEOUT_S": "3"})
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert 2.5 <= dt < 60.0
    assert "B9_LAUNCH_TIMEOUT_S" in r.stderr
    assert _dist_dirs() == before, "the launcher left its bootstrap directory behind"


def test_whole_run_deadline():
    r, dt = _run([EXE, "--gpus", "1", "--forceRanks"], {"B9_TEST_STALL": "start:0", "B9_LAUNCH_TIMEOUT_S": "0", "B9_RUN_TIMEOUT_S": "2"})
    assert r.returncode == 124 and "B9_RUN_TIMEOUT_S" in r.stderr and dt < 60.0


@pytest.mark.parametrize("which", ["cli", "bench"])
def test_exit_code_is_the_first_failure_not_the_sigterm_of_its_peers(which):
    """Two ranks; rank 0 fails on its own (no model directory / no GPU here) while rank 1 is parked: the launcher ends rank 1
    with SIGTERM and reports rank 0's exit code -- not 143 (or 15), the status of the rank it killed itself."""
    before = _dist_dirs()
    cmd = [EXE, "--gpus", "2"] if which == "cli" else [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"]
    r, dt = _run(cmd, {"B9_TEST_STALL": "start:1", "B9_LAUNCH_TIMEOUT_S": "100"}, timeout=300)
    assert r.returncode not in (0, 143, 15, 124), (r.returncode, r.stderr[-2000:])
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    assert dt < 100.0
    assert _dist_dirs() == before


@pytest.mark.parametrize("which", ["cli", "bench"])
def test_sigterm_to_the_launcher_ends_its_ranks_and_cleans_up(which):
    """SIGTERM (or ^C) to the launcher is forwarded to the ranks, they are reaped, the bootstrap directory is
    removed, and the exit code says which signal it was -- no rank is left parked in ncclCommInitRank."""
    import signal
    before = _dist_dirs()
    cmd = [EXE, "--gpus", "1", "--forceRanks"] if which == "cli" else [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-ranks"]
    env = dict(os.environ, B9_TEST_STALL="start:0", B9_LAUNCH_TIMEOUT_S="100")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "B9_RANK"):
        env.pop(k, None)
    env["B9_TEST_STALL"] = "start:0"
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.monotonic() + 60.0
    kids = []
    while time.monotonic() < deadline and len(kids) < 1:      # the rank has started (a child of the launcher)
        kids = [int(x) for x in subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()]
        time.sleep(0.05)
    assert len(kids) == 1, kids
    time.sleep(1.0)                                           # (parked in the test hook by now)
    t0 = time.monotonic()
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert time.monotonic() - t0 < 20.0
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-2000:])
    assert "signal 15" in err
    for k in kids:                                           # reaped, not orphaned
        assert not os.path.exists(f"/proc/{k}"), k
    assert _dist_dirs() == before, "the launcher left its bootstrap directory behind"


def test_rccl_bootstrap_refuses_a_launch_without_an_id(tmp_path):
    """ranks started by a launcher that names no launch (only RANK exported) would all use one id-file name, so
    a rank > 0 could read a dead attempt's id and hang: the exchange refuses before touching a GPU and says what to export."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from base_amd import hostlib\n"
            "try:\n"
            "    hostlib.Exchange.rccl(1, 2, 0, %r)\n"
            "except Exception as e:\n"
            "    print('REFUSED', e); sys.exit(0)\n"
            "sys.exit(3)\n") % (ROOT, str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k not in ("B9_LAUNCH_NONCE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "REFUSED" in r.stdout and "B9_LAUNCH_NONCE" in r.stdout, (r.returncode, r.stdout, r.stderr[-1500:])


def test_a_multi_rank_line_needs_distinct_gpus():
    """two ranks that report the same PCI bus id (or a communicator with another rank count) make
    bench.py exit non-zero instead of printing a line that looks like a 2-GPU result; the C++ exchange applies the same
    check to itself (b9h::group_error)."""
    import ctypes as C
    from base_amd import hostlib
    lib = hostlib.load()
    msg = C.create_string_buffer(512)
    assert lib.b9h_group_check(2, 2, b"0000:05:00.0,0000:15:00.0", msg, 512) == 0 and msg.value == b""
    assert lib.b9h_group_check(2, 2, b"0000:05:00.0,0000:05:00.0", msg, 512) == 1 and b"share the GPU 0000:05:00.0" in msg.value
    assert lib.b9h_group_check(2, 1, b"0000:05:00.0,0000:15:00.0", msg, 512) == 1 and b"1 rank(s)" in msg.value
    assert lib.b9h_group_check(4, 4, b"a,b,c", msg, 512) == 1
    assert lib.b9h_group_check(1, 0, b"", msg, 512) == 0                      # one rank: nothing to check
    code = "import sys; sys.path.insert(0, %r); import bench; bench.check_group(2, 2, ['0000:05:00.0', '0000:05:00.0'])" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "share the GPU" in r.stderr, (r.returncode, r.stderr[-1000:])
    code = "import sys; sys.path.insert(0, %r); import bench; bench.check_group(2, 2, ['0000:05:00.0', '0000:15:00.0'])" % ROOT
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).returncode == 0


def test_a_profile_of_other_sources_drops_the_counters(tmp_path, monkeypatch):
    """bench.py divides the profile's per-launch counters by the LIVE launch time: when the kernel sources have changed
    since the profile was taken, the counters are dropped (with the reason) instead of silently mixing two builds."""
    sys.path.insert(0, ROOT)
    import bench
    from base_amd import build
    h = build.source_hash()
    doc = {"commit": "abc1234", "csrc_sha256": h, "command": "x", "kernels": {"k_mcmc_step<8, 1>": {"avg_us": 15.0}},
           "pmc": {"k_mcmc_step<8, 1>": {"hbm_bytes_per_launch": 1.0e7, "SQ_INSTS_VALU": 4.0e6, "SQ_ACTIVE_INST_VALU": 4.0e6,
                                        "SQ_WAVE_CYCLES": 1.9e7, "SQ_WAVES": 2976}}}
    os.makedirs(tmp_path / "profiles")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    pth = tmp_path / "profiles" / f"{bench.PROFILE_TAG}_summary.json"
    pth.write_text(json.dumps(doc))
    fresh = bench.profile_counters("k_mcmc_step", h)
    assert "stale" not in fresh and fresh["valu_active_quad_cycles_per_launch"] == 4.0e6
    roof = bench.valu_roofline(fresh, 15e-6, 1.0e8)
    assert 0.4 < roof["frac"] < 0.5 and roof["traffic"] == 1.0e7 and 0.1 < roof["useful_frac"] < 0.3
    # the same file against a tree whose sources differ
    stale = bench.profile_counters("k_mcmc_step", "0" * 64)
    assert "stale" in stale and "other kernel sources" in stale["stale"]
    roof = bench.valu_roofline(stale, 15e-6, 1.0e8)
    assert roof["frac"] is None and roof["achieved"] is None and roof["traffic"] is None and roof["hbm"]["frac"] is None
    assert roof["useful_frac"] is not None            # needs no counter
    pth.unlink()
    assert "stale" in bench.profile_counters("k_mcmc_step", h)


def test_useful_operation_model():
    sys.path.insert(0, ROOT)
    import bench
    ops = bench.useful_lane_ops(8, 400, 0.3)
    assert ops["search_rounds"] == 3 and ops["single"] < ops["binary"]
    assert abs(ops["mean"] - (0.7 * ops["single"] + 0.3 * ops["binary"])) < 1e-9
    assert bench.useful_lane_ops(8, 400, 0.3, n_pops=2)["single"] > ops["single"]


def _write_part(path, n_rows, first_walker, per, comment=True):
    with open(path, "w") as f:
        if comment:
            f.write("# base9_hip ABI 3; mode=givenMass; populations=1; walkers=4\n")
        f.write("      logAge      logPost stage\n")
        for k in range(n_rows):
            f.write(f"{9.0 + 0.001 * (k // per):12.6f} {-100.0 - (first_walker + k % per):14.6f} {3:5d}\n")


def test_part_file_merge_is_checked(tmp_path):
    """Rank 0's merge after a --gpus N run: rows interleaved in walker order; a truncated part or an unwritable
    output is an ERROR that keeps every part file and leaves no half-merged .res behind."""
    from base_amd import hostlib
    lib = hostlib.load()
    base = str(tmp_path / "run.res")
    per, steps = 2, 5
    for r in range(2):
        _write_part(f"{base}.part{r}", per * steps, r * per, per)
    assert lib.b9h_merge_parts(base.encode(), 2, per, per * steps) == 0
    lines = open(base).read().splitlines()
    assert lines[0].startswith("# base9_hip ABI 3") and lines[1].split()[0] == "logAge" and len(lines) == 2 + 2 * per * steps
    walkers = [int(round(-float(ln.split()[1]) - 100.0)) for ln in lines[2:]]
    assert walkers == [0, 1, 2, 3] * steps                      # walker order within every step
    assert not os.path.exists(base + ".part0") and not os.path.exists(base + ".part1")
    # a truncated part: error, parts kept, no merged file
    os.unlink(base)
    _write_part(f"{base}.part0", per * steps, 0, per)
    _write_part(f"{base}.part1", per * steps - 3, per, per)
    assert lib.b9h_merge_parts(base.encode(), 2, per, per * steps) != 0
    assert b"part1 holds 7 rows, expected 10" in lib.b9h_last_error()
    assert os.path.exists(base + ".part0") and os.path.exists(base + ".part1") and not os.path.exists(base)
    # a part with rows left over
    _write_part(f"{base}.part1", per * steps + 2, per, per)
    assert lib.b9h_merge_parts(base.encode(), 2, per, per * steps) != 0
    assert os.path.exists(base + ".part0") and not os.path.exists(base)
    # an output that cannot be written
    _write_part(f"{base}.part1", per * steps, per, per)
    bad = str(tmp_path / "no_such_dir" / "run.res")
    os.makedirs(os.path.dirname(bad))
    for r in range(2):
        os.replace(f"{base}.part{r}", f"{bad}.part{r}")
    os.chmod(os.path.dirname(bad), 0o500)
    try:
        if os.geteuid() != 0:                                     # (root writes anywhere)
            assert lib.b9h_merge_parts(bad.encode(), 2, per, per * steps) != 0
            assert os.path.exists(bad + ".part0") and os.path.exists(bad + ".part1")
    finally:
        os.chmod(os.path.dirname(bad), 0o700)
