"""Host side (C++, base_amd/host/): the model-pack / photometry / settings parsers round-trip what
base_amd.synth writes (CPU, through libbase9host.so); the CLI programs run on the GPU and their
outputs agree with the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle
from base_amd import abi, host_build, synth
from conftest import build_problem

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "base_amd", "host")


def _res_head(path):
    """(the provenance note of the sidecar <path>.meta, the column names) of a .res file -- which by default starts with
    its ONE header line, the [RECALL] layout (a consumer that skips exactly one line reads data next)."""
    with open(path) as f:
        first = f.readline()
    assert not first.startswith("#"), first
    note = open(path + ".meta").read()
    assert note.startswith("base9_hip ABI "), note
    return note, first.split()


@pytest.fixture(scope="module")
def hostlib():
    from base_amd import build
    build.build_hip()
    host_build.build_host()
    lib = C.CDLL(os.path.join(HOST, "libbase9host.so"))
    lib.b9h_last_error.restype = C.c_char_p
    lib.b9h_load_pack.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(abi.b9_pack)]
    lib.b9h_free_pack.argtypes = [C.c_void_p]
    lib.b9h_read_phot.argtypes = [C.c_char_p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_void_p), C.POINTER(abi.b9_stars), C.c_char_p, C.c_int]
    lib.b9h_free_phot.argtypes = [C.c_void_p]
    lib.b9h_settings_dump.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
    return lib


def _arr(ptr, n, dtype=np.float64):
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy() if n else np.zeros(0, dtype)


@pytest.mark.parametrize("name,n_filt,n_y,wd_ragged", [("parsec", 8, 1, False), ("dsed", 5, 3, True)])
def test_model_pack_round_trip(hostlib, tmp_path, name, n_filt, n_y, wd_ragged):
    pack_d = synth.make_pack(name, n_filt=n_filt, n_y=n_y, n_feh=3, n_age=4, n_eep=30, wd_ragged=wd_ragged)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    # ask for the filters in a different order than the files hold them
    want = list(reversed(pack_d["filters"]))
    h, view = C.c_void_p(), abi.b9_pack()
    rc = hostlib.b9h_load_pack(root.encode(), name.encode(), b"montgomery", ",".join(want).encode(), C.byref(h), C.byref(view))
    assert rc == 0, hostlib.b9h_last_error()
    try:
        assert (view.n_filt, view.n_feh, view.n_y, view.n_age) == (n_filt, 3, n_y, 4)
        np.testing.assert_array_equal(_arr(view.feh, 3), pack_d["feh"])
        np.testing.assert_array_equal(_arr(view.log_age, 4), pack_d["log_age"])
        n_iso = 3 * n_y * 4
        np.testing.assert_array_equal(_arr(view.iso_first_eep, n_iso, np.int32), pack_d["iso_first_eep"])
        np.testing.assert_array_equal(_arr(view.iso_n_eep, n_iso, np.int32), pack_d["iso_n_eep"])
        np.testing.assert_array_equal(_arr(view.iso_offset, n_iso, np.int64), pack_d["iso_offset"])
        np.testing.assert_array_equal(_arr(view.mass, view.n_points), pack_d["mass"])
        np.testing.assert_array_equal(_arr(view.mags, view.n_points * n_filt).reshape(-1, n_filt), pack_d["mags"][:, ::-1])
        np.testing.assert_array_equal(_arr(view.abs_coeff, n_filt), pack_d["abs_coeff"][::-1])
        # cooling tracks: one per (carbonicity, mass) node, each with its own age axis (rectangular packs repeat theirs)
        tracks = synth.wd_cooling_tracks(pack_d)
        n_tracks = view.n_wc_carb * view.n_wc_mass
        assert n_tracks == len(tracks) and view.n_wc_points == sum(len(t[0]) for t in tracks)
        n_age, off = _arr(view.wc_n_age, n_tracks, np.int32), _arr(view.wc_offset, n_tracks, np.int64)
        age, te, ra = (_arr(p, view.n_wc_points) for p in (view.wc_log_age, view.wc_log_teff, view.wc_log_radius))
        for t, (a, b, c) in enumerate(tracks):
            assert n_age[t] == len(a)
            np.testing.assert_array_equal(age[off[t]:off[t] + n_age[t]], a)
            np.testing.assert_array_equal(te[off[t]:off[t] + n_age[t]], b)
            np.testing.assert_array_equal(ra[off[t]:off[t] + n_age[t]], c)
        if wd_ragged:
            assert len(set(n_age.tolist())) > 1
        at = pack_d["at_mags"].reshape(2, view.n_at_logg, view.n_at_teff, n_filt)[..., ::-1]
        np.testing.assert_array_equal(_arr(view.at_mags, at.size), at.ravel())
        assert view.n_at_type == 2
    finally:
        hostlib.b9h_free_pack(h)
    # a filter the model does not provide is an error, not a silent column of zeros
    rc = hostlib.b9h_load_pack(root.encode(), name.encode(), b"montgomery", b"U,Zz", C.byref(h), C.byref(view))
    assert rc != 0 and b"Zz" in hostlib.b9h_last_error()


def test_photometry_round_trip_and_magnitude_window(hostlib, tmp_path):
    pack_d, cl, *_ = build_problem("parsec", 8, n_stars=120, wd_frac=0.1)
    path = synth.write_phot(cl, pack_d["filters"], str(tmp_path / "c.phot"))
    h, view, buf = C.c_void_p(), abi.b9_stars(), C.create_string_buffer(256)
    assert hostlib.b9h_read_phot(path.encode(), -1e300, 1e300, 0, C.byref(h), C.byref(view), buf, 256) == 0
    assert buf.value.decode().split(",") == list(pack_d["filters"]) and view.n_stars == 120
    np.testing.assert_array_equal(_arr(view.obs, 120 * 8).reshape(120, 8), cl["obs"])
    np.testing.assert_array_equal(_arr(view.sigma, 120 * 8).reshape(120, 8), cl["sigma"])
    np.testing.assert_array_equal(_arr(view.mass1, 120), cl["mass1"])
    np.testing.assert_array_equal(_arr(view.clust_prior, 120), cl["clust_prior"])
    np.testing.assert_array_equal(_arr(view.stage, 120, np.int32), cl["stage"])
    np.testing.assert_array_equal(_arr(view.wd_type, 120, np.int32), cl["wd_type"])
    hostlib.b9h_free_phot(h)
    # magnitude window in filter 2 drops MS stars outside it but keeps WDs
    v = cl["obs"][:, 2]
    lo, hi = np.percentile(v, 20), np.percentile(v, 80)
    assert hostlib.b9h_read_phot(path.encode(), lo, hi, 2, C.byref(h), C.byref(view), buf, 256) == 0
    keep = ((v >= lo) & (v <= hi)) | (cl["stage"] == abi.STAGE_WD)
    assert view.n_stars == keep.sum()
    hostlib.b9h_free_phot(h)
    bad = tmp_path / "bad.phot"
    bad.write_text("id U B sigU mass1 massRatio stage CMprior useDBI\n1 1 2 0.1 1 0 1 0.9 1\n")
    assert hostlib.b9h_read_phot(str(bad).encode(), -1e300, 1e300, 0, C.byref(h), C.byref(view), buf, 256) != 0


def test_settings_yaml_and_flag_override(hostlib, tmp_path):
    pack_d, cl, *_ = build_problem("dsed", 3, n_stars=10)
    y = synth.write_yaml(str(tmp_path / "base9.yaml"), "a.phot", "models", "out", cl["truth"], ms_model="dsed")
    args = [b"prog", b"--config", y.encode(), b"--sigmaFe_H", b"0.05", b"--photFile=b.phot", b"--runIter", b"123"]
    argv = (C.c_char_p * len(args))(*args)
    out = C.create_string_buffer(8192)
    assert hostlib.b9h_settings_dump(len(args), argv, out, 8192) == 0, hostlib.b9h_last_error()
    kv = dict(l.split(" = ", 1) for l in out.value.decode().strip().split("\n"))
    assert kv["general.files.photFile"] == "b.phot" and kv["general.files.modelDirectory"] == "models"
    assert kv["general.cluster.priors.sigmas.Fe_H"] == "0.05" and kv["general.cluster.priors.sigmas.distMod"] == "0.3"
    assert kv["singlePopMcmc.runIter"] == "123" and kv["general.main_sequence.msRgbModel"] == "dsed"
    assert float(kv["general.cluster.starting.logAge"]) == cl["truth"][abi.P_LOGAGE]
    args = [b"prog", b"--noSuchFlag", b"1"]
    assert hostlib.b9h_settings_dump(2, (C.c_char_p * 3)(*args), out, 8192) != 0
    # the value-less switches of this build: --resComment (the provenance note inside the .res too), --forceRanks, --marginalise
    args = [b"prog", b"--resComment", b"--forceRanks", b"--marginalise", b"--walkers", b"4"]
    assert hostlib.b9h_settings_dump(len(args), (C.c_char_p * len(args))(*args), out, 8192) == 0, hostlib.b9h_last_error()
    kv = dict(l.split(" = ", 1) for l in out.value.decode().strip().split("\n"))
    assert kv["gpu.resComment"] == "1" and kv["gpu.forceRanks"] == "1" and kv["gpu.marginalise"] == "1" and kv["gpu.walkers"] == "4"


def _cli(name, *args):
    exe = os.path.join(HOST, "bin", name)
    return subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=600)


@pytest.mark.gpu
def test_makecmd_matches_oracle(hostlib, tmp_path):
    pack_d = synth.make_pack("parsec", 8, n_feh=4, n_age=6, n_eep=80)
    truth = synth.default_params(pack_d)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    y = synth.write_yaml(str(tmp_path / "base9.yaml"), "unused.phot", root, str(tmp_path / "iso"), truth)
    r = _cli("makeCMD", "--config", y)
    assert r.returncode == 0, r.stderr
    rows = np.loadtxt(str(tmp_path / "iso.cmd"), skiprows=2)
    first, mass, mags, tip = oracle.derive_isochrone(oracle.load(), abi.make_pack(pack_d), truth)
    assert rows.shape == (len(mass), 10) and int(rows[0, 0]) == first
    np.testing.assert_allclose(rows[:, 1], mass, rtol=0, atol=1e-10)
    app = mags + truth[abi.P_MOD] + (pack_d["abs_coeff"] - 1.0) * truth[abi.P_ABS]
    np.testing.assert_allclose(rows[:, 2:], app, rtol=0, atol=1e-8)
    r = _cli("makeCMD", "--config", y, "--logAge", "3.0")
    assert r.returncode != 0 and "outside the model grid" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prog,n_pops,n_y", [("singlePopMcmc", 1, 1), ("multiPopMcmc", 2, 3)])
def test_mcmc_cli_runs_and_recovers_truth(hostlib, tmp_path, prog, n_pops, n_y):
    pack_d = synth.make_pack("dsed", 8, n_y=n_y, n_feh=4, n_age=8, n_eep=90)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, 1500, seed=21, truth=truth, wd_frac=0.03, n_pops=n_pops)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    phot = synth.write_phot(cl, pack_d["filters"], str(tmp_path / "c.phot"))
    start = truth.copy()                                  # start AWAY from the truth: the chain has to find it
    start[abi.P_LOGAGE] += 0.008; start[abi.P_MOD] += 0.015; start[abi.P_FEH] -= 0.02
    y = synth.write_yaml(str(tmp_path / "base9.yaml"), phot, root, str(tmp_path / "run"), start, ms_model="dsed",
                         burn=4000, run=1500, walkers=4)
    extra = ["--priorFe_H", repr(float(truth[abi.P_FEH])), "--priorDistMod", repr(float(truth[abi.P_MOD])),
             "--priorAv", repr(float(truth[abi.P_ABS]))]
    if n_pops == 2:
        extra += ["--startingYA", repr(float(truth[abi.P_Y])), "--startingYB", repr(float(truth[abi.P_Y2])), "--startingLambda", "0.5"]
    r = _cli(prog, "--config", y, *extra)
    assert r.returncode == 0, r.stderr
    assert "star-likelihood evals/s" in r.stderr
    note, head = _res_head(str(tmp_path / "run.res"))
    res = np.loadtxt(str(tmp_path / "run.res"), skiprows=1)
    assert "mode=givenMass" in note and f"populations={n_pops}" in note and "walkers=4" in note
    assert head[0] == "logAge" and head[-2:] == ["logPost", "stage"] and res.shape == (5500 * 4, len(head))
    main = res[res[:, -1] == 3]
    assert len(main) == 1500 * 4 and np.all(np.isfinite(main[:, -2]))
    col = {n: i for i, n in enumerate(head)}
    acc = float(r.stderr.split("acceptance")[1].split()[0])
    assert 0.05 < acc < 0.7, acc
    # the recorded log-posterior of the last row is what the oracle gives at that position
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    cl2 = dict(cl)
    lo, hi = np.where(np.asarray(cl["sigma"]) > 0, cl["obs"], np.inf).min(axis=0), np.where(np.asarray(cl["sigma"]) > 0, cl["obs"], -np.inf).max(axis=0)
    cl2["filter_prior_min"], cl2["filter_prior_max"] = lo, hi          # the reader's field-star box
    pri = synth.default_priors(pack_d, truth, n_pops)
    for k, v in ((abi.P_Y, 0.0), (abi.P_Y2, 0.0)):
        pri.var[k] = v
    row = truth.copy()
    for name, i in col.items():
        key = {"logAge": abi.P_LOGAGE, "FeH": abi.P_FEH, "modulus": abi.P_MOD, "absorption": abi.P_ABS, "Y": abi.P_Y,
               "YA": abi.P_Y, "YB": abi.P_Y2, "lambda": abi.P_LAMBDA}.get(name)
        if key is not None:
            row[key] = main[-1, i]
    row[abi.P_IFMR_INTERCEPT], row[abi.P_IFMR_SLOPE], row[abi.P_IFMR_QUAD] = 0.77, 0.08, 0.0
    orc = oracle.Oracle(pack, abi.make_stars(cl2), pri, abi.make_options(n_pops=n_pops))
    want = orc.logpost(row[None, :])[0]
    assert abs(main[-1, -2] - want) <= 2e-4 * max(1.0, abs(want))     # .res holds 6 decimals of each parameter
    # Convergence, judged on the posterior itself (the synthetic grids are coarse, so FeH/age/modulus lie
    # along a long near-degenerate ridge and "mean == truth" is not a fair test of a short chain): the
    # start was far down the slope, the main run sits at least as high as the truth does.
    t_row, s_row = row.copy(), row.copy()
    for k in (abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS, abi.P_Y, abi.P_Y2, abi.P_LAMBDA):
        t_row[k] = truth[k]; s_row[k] = truth[k]
    for k in (abi.P_LOGAGE, abi.P_FEH, abi.P_MOD):
        s_row[k] = start[k]
    if n_pops == 2:
        t_row[abi.P_LAMBDA] = s_row[abi.P_LAMBDA] = 0.5
    lp_truth, lp_start = orc.logpost(np.stack([t_row, s_row]))
    assert lp_start < lp_truth - 50.0, (lp_start, lp_truth)
    assert main[:, -2].mean() > lp_truth - (6.0 + 0.5 * (len(head) - 2)), (main[:, -2].mean(), lp_truth)
    assert np.linalg.norm(main[:, col["logAge"]].mean() - truth[abi.P_LOGAGE]) < abs(start[abi.P_LOGAGE] - truth[abi.P_LOGAGE])


@pytest.mark.gpu
def test_samplemass_cli_matches_oracle(hostlib, tmp_path):
    """singlePopMcmc -> sampleMass: the .massSamples / .membership rows are the oracle's draws for the chain's
    main-run rows (row index = position among them, seed = general.seed)."""
    pack_d = synth.make_pack("parsec", 4, n_feh=4, n_age=6, n_eep=60)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, 120, seed=5, truth=truth, wd_frac=0.05)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    phot = synth.write_phot(cl, pack_d["filters"], str(tmp_path / "c.phot"))
    y = synth.write_yaml(str(tmp_path / "base9.yaml"), phot, root, str(tmp_path / "run"), truth, burn=200, run=30, walkers=2)
    assert _cli("singlePopMcmc", "--config", y).returncode == 0
    r = _cli("sampleMass", "--config", y, "--margIsoIncrem", "2", "--nMassRatios", "3", "--seed", "31")
    assert r.returncode == 0, r.stderr
    assert "star draws/s" in r.stderr
    note, head = _res_head(str(tmp_path / "run.res"))
    res = np.loadtxt(str(tmp_path / "run.res"), skiprows=1)
    main = res[res[:, -1] == 3]
    ms = np.loadtxt(str(tmp_path / "run.massSamples"), skiprows=1)
    mb = np.loadtxt(str(tmp_path / "run.membership"), skiprows=1)
    assert ms.shape == (len(main), 2 * 120) and mb.shape == (len(main), 120) and len(main) == 60
    ids = open(str(tmp_path / "run.membership")).readline().split()
    assert ids[:3] == ["1", "2", "3"]
    rows = np.tile(truth, (len(main), 1))
    for name, i in {n: i for i, n in enumerate(head)}.items():
        key = {"logAge": abi.P_LOGAGE, "FeH": abi.P_FEH, "modulus": abi.P_MOD, "absorption": abi.P_ABS}.get(name)
        if key is not None:
            rows[:, key] = main[:, i]
    rows[:, abi.P_IFMR_INTERCEPT], rows[:, abi.P_IFMR_SLOPE], rows[:, abi.P_IFMR_QUAD] = 0.77, 0.08, 0.0
    cl2 = dict(cl)
    sg = np.asarray(cl["sigma"])
    cl2["filter_prior_min"] = np.where(sg > 0, cl["obs"], np.inf).min(axis=0)          # the reader's field-star box
    cl2["filter_prior_max"] = np.where(sg > 0, cl["obs"], -np.inf).max(axis=0)
    opt = abi.make_options(marg_iso_increm=2, marg_n_q=3)
    om, oq, omem, _, margin = oracle.Oracle(abi.make_pack(pack_d), abi.make_stars(cl2), synth.default_priors(pack_d, truth), opt).sample_mass(rows, seed=31)
    safe = margin > 1e-4                      # the .res file holds 6 decimals of each parameter: near-ties may flip
    assert safe.mean() > 0.98
    assert np.max(np.abs(ms[:, 0::2][safe] - om[safe])) < 2e-6 and np.max(np.abs(ms[:, 1::2][safe] - oq[safe])) < 1e-4
    assert np.max(np.abs(mb - omem)) < 1e-4


def test_corrupted_input_files_are_rejected_or_read_never_crash(hostlib, tmp_path):
    """Random damage to the photometry file and to every model file (truncation, cut lines, junk tokens and
    lines, blanked lines, shuffled tokens): the readers either accept what is still a well-formed file or
    return an error with a message."""
    import random, shutil
    pack_d = synth.make_pack("dsed", 4, n_feh=3, n_age=4, n_eep=30)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, 40, seed=3, truth=truth, wd_frac=0.1)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    phot = synth.write_phot(cl, pack_d["filters"], str(tmp_path / "c.phot"))
    filters = ",".join(pack_d["filters"]).encode()
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs]
    rng = random.Random(7)

    def corrupt(path, out):
        lines = open(path).read().split("\n")
        k, i = rng.randrange(6), rng.randrange(len(lines))
        if k == 0: lines = lines[:i]
        elif k == 1: lines[i] = lines[i][: len(lines[i]) // 2]
        elif k == 2: lines[i] = lines[i].replace(" ", " x ", 1)
        elif k == 3: lines.insert(i, "nan inf -1e999 &&&")
        elif k == 4: lines[i] = ""
        else: lines[i] = " ".join(reversed(lines[i].split()))
        open(out, "w").write("\n".join(lines))

    rejected = 0
    for _ in range(80):
        out = str(tmp_path / "f.phot"); corrupt(phot, out)
        h, view, buf = C.c_void_p(), abi.b9_stars(), C.create_string_buffer(4096)
        rc = hostlib.b9h_read_phot(out.encode(), -1e300, 1e300, 0, C.byref(h), C.byref(view), buf, 4096)
        if rc == 0: hostlib.b9h_free_phot(h)
        else: rejected += 1; assert hostlib.b9h_last_error()
        src = rng.choice(files)
        r2 = str(tmp_path / "m2"); shutil.rmtree(r2, ignore_errors=True); shutil.copytree(root, r2)
        corrupt(src, os.path.join(r2, os.path.relpath(src, root)))
        h, pv = C.c_void_p(), abi.b9_pack()
        rc = hostlib.b9h_load_pack(r2.encode(), b"dsed", b"montgomery", filters, C.byref(h), C.byref(pv))
        if rc == 0: hostlib.b9h_free_pack(h)
        else: rejected += 1; assert hostlib.b9h_last_error()
    assert rejected > 40


@pytest.mark.gpu
def test_mcmc_cli_marginalised_mode(hostlib, tmp_path):
    """--marginalise: every star integrated over primary mass and mass ratio (b9_options.mode), so the catalogue's mass
    columns are hints only -- a .phot whose mass1 column is 0 runs, and the recorded log-posterior is the oracle's
    marginalised one.  (In given-mass mode the same file is refused with a message.)"""
    pack_d = synth.make_pack("dsed", 8, n_feh=4, n_age=8, n_eep=60)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, 120, seed=3, truth=truth)
    cl0 = dict(cl); cl0["mass1"] = np.zeros_like(cl["mass1"]); cl0["mass_ratio"] = np.zeros_like(cl["mass_ratio"])
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    phot = synth.write_phot(cl0, pack_d["filters"], str(tmp_path / "c.phot"))
    y = synth.write_yaml(str(tmp_path / "base9.yaml"), phot, root, str(tmp_path / "run"), truth, ms_model="dsed", burn=20, run=10, walkers=2)
    r = _cli("singlePopMcmc", "--config", y)
    assert r.returncode != 0 and "mass1 > 0" in r.stderr
    r = _cli("singlePopMcmc", "--config", y, "--marginalise", "--margIsoIncrem", "2", "--nMassRatios", "2", "--block", "10")
    assert r.returncode == 0, r.stderr
    assert "marginalised mode" in r.stderr
    note, head = _res_head(str(tmp_path / "run.res"))
    assert "mode=marginalised (margIsoIncrem=2, nMassRatios=2)" in note and "walkers=2" in note
    res = np.loadtxt(str(tmp_path / "run.res"), skiprows=1)
    assert res.shape == (30 * 2, len(head))
    # --resComment: the same note as a leading "# ..." line of the .res itself, nothing else changes
    plain = open(str(tmp_path / "run.res")).read()
    r = _cli("singlePopMcmc", "--config", y, "--marginalise", "--margIsoIncrem", "2", "--nMassRatios", "2", "--block", "10", "--resComment")
    assert r.returncode == 0, r.stderr
    noted = open(str(tmp_path / "run.res")).read()
    assert noted == "# " + note + plain
    cl2 = dict(cl0)
    lo, hi = np.where(np.asarray(cl["sigma"]) > 0, cl["obs"], np.inf).min(axis=0), np.where(np.asarray(cl["sigma"]) > 0, cl["obs"], -np.inf).max(axis=0)
    cl2["filter_prior_min"], cl2["filter_prior_max"] = lo, hi
    pri = synth.default_priors(pack_d, truth, 1)
    for k in (abi.P_Y, abi.P_Y2):
        pri.var[k] = 0.0
    row = truth.copy()
    for name, i in {n: i for i, n in enumerate(head)}.items():
        if name in abi.PARAM_NAMES:
            row[abi.PARAM_NAMES.index(name)] = res[-1, i]
    opt = abi.make_options(abi.MODE_MARGINALISED, 1, 2, 2)
    want = oracle.Oracle(abi.make_pack(pack_d), abi.make_stars(cl2), pri, opt).logpost(row[None, :])[0]
    # the .res holds 6 decimals of each parameter: compare at the rounded position, allowing for the posterior's slope
    # over half a unit of the last place (|dlogPost/dlogAge| ~ 1e3-1e4 here)
    assert abs(res[-1, -2] - want) <= 1e-2, (res[-1, -2], want)


def test_mcmc_cli_without_a_gpu_fails_loudly_and_at_once(hostlib, tmp_path):
    """The product has no CPU fallback: on a box without a HIP device the sampler CLI says so and exits non-zero -- also
    as `--gpus 2`, where the launcher must bring every rank down promptly instead of leaving one blocked.  (Skipped where a
    GPU is present: there the same command simply runs.)"""
    import time
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    pack_d = synth.make_pack("dsed", 8, n_feh=4, n_age=8, n_eep=90)
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, 60, seed=2, truth=truth)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    phot = synth.write_phot(cl, pack_d["filters"], str(tmp_path / "c.phot"))
    y = synth.write_yaml(str(tmp_path / "base9.yaml"), phot, root, str(tmp_path / "run"), truth, ms_model="dsed", burn=20, run=10, walkers=4)
    for extra in ([], ["--gpus", "2"]):
        t0 = time.time()
        r = _cli("singlePopMcmc", "--config", y, *extra)
        assert r.returncode != 0, r.stdout
        assert "no HIP device" in r.stderr or "hip" in r.stderr.lower(), r.stderr
        assert time.time() - t0 < 60
        assert not os.path.exists(str(tmp_path / "run.res"))
