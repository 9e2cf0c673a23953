"""GPU edge cases of the C ABI: filter counts up to the 16-filter limit, long isochrones, call-order
and argument errors, reloading packs, degenerate inputs."""
import ctypes as C

import numpy as np
import pytest

import oracle
from base_amd import abi, synth
from conftest import build_problem

pytestmark = pytest.mark.gpu


def _err(a, b):
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    return float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin])))) if fin.any() else 0.0


@pytest.mark.parametrize("n_filt", [1, 2, 4, 9, 12, 16])
def test_filter_counts(n_filt):
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", n_filt, n_stars=300, wd_frac=0.05)
    eng = engine.Engine(pack, stars, priors, options)
    params = synth.walker_params(cl["truth"], 3)
    got = eng.logpost(params, perstar=True)
    want = oracle.Oracle(pack, stars, priors, options).logpost(params, perstar=True)
    assert _err(got[1], want[1]) <= 1e-9 and _err(got[0], want[0]) <= 1e-9
    assert eng.bytes_per_star_eval() == 16 * n_filt + 36


def test_seventeen_filters_rejected():
    from base_amd import engine
    pack_d = synth.make_pack("parsec", 17, n_feh=3, n_age=4, n_eep=30)
    with pytest.raises(engine.B9Error) as e:
        engine.Engine(abi.make_pack(pack_d))
    assert e.value.code == abi.B9_ERR_CAPACITY


def test_long_isochrone_2000_eeps():
    """2000 EEPs x 8 filters: mass column 16 KB in LDS, marginalised-mode isochrone 144 KB in LDS."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=400, n_feh=3, n_age=4, n_eep=2000)
    eng = engine.Engine(pack, stars, priors, options)
    params = synth.walker_params(cl["truth"], 2)
    assert _err(eng.logpost(params, perstar=True)[1], oracle.Oracle(pack, stars, priors, options).logpost(params, perstar=True)[1]) <= 1e-9
    assert eng.max_eep() >= 1996
    sub = {k: (np.asarray(v)[:6] if k in ("obs", "sigma", "mass1", "mass_ratio", "clust_prior", "stage", "wd_type") else v) for k, v in cl.items()}
    mopt = abi.make_options(abi.MODE_MARGINALISED, 1, 1, 2)
    s6 = abi.make_stars(sub)
    m_eng = engine.Engine(pack, s6, priors, mopt)
    assert _err(m_eng.logpost(params), oracle.Oracle(pack, s6, priors, mopt).logpost(params)) <= 1e-9

    # the fused sampler step on the same long isochrones: the hot role stages 2 x 2000 masses per workgroup
    # (past the 4 x 256 elements that travel through registers) and every candidate takes 71 derivation parts
    from base_amd import mcmc
    free, chol = np.array(mcmc.DEFAULT_FREE), np.diag([3e-4, 2e-3, 8e-4, 6e-4])
    start = synth.walker_params(cl["truth"], 3, seed=4, scale=0.1)
    lp0 = eng.logpost(start)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, np.arange(3), free, chol, 9, 0, 8)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, np.arange(3), free, chol, 9, 0, 8)
    assert dev[4] == host[4]
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)


def test_call_order_and_argument_errors():
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=50)
    eng = engine.Engine()
    with pytest.raises(engine.B9Error) as e:
        eng.n_stars = 50
        eng.logpost(cl["truth"][None, :])
    assert e.value.code == abi.B9_ERR_STATE
    eng.load_pack(pack)
    with pytest.raises(engine.B9Error) as e:
        eng.logpost(cl["truth"][None, :])
    assert e.value.code == abi.B9_ERR_STATE
    # stars with the wrong filter count are refused at first use
    other = synth.make_cluster(synth.make_pack("dsed", 5, n_feh=4, n_age=8, n_eep=90), 20, seed=1)
    eng.load_stars(abi.make_stars(other))
    with pytest.raises(engine.B9Error) as e:
        eng.logpost(cl["truth"][None, :])
    assert e.value.code == abi.B9_ERR_INVALID
    eng.load_stars(stars)
    assert np.isfinite(eng.logpost(cl["truth"][None, :])[0])
    # malformed inputs
    bad = dict(pack_d); bad["feh"] = pack_d["feh"][::-1].copy()
    with pytest.raises(engine.B9Error):
        engine.Engine(abi.make_pack(bad))
    badc = dict(cl); badc["clust_prior"] = np.zeros(50)
    with pytest.raises(engine.B9Error):
        engine.Engine(pack, abi.make_stars(badc))
    with pytest.raises(engine.B9Error):
        eng.set_options(abi.make_options(n_pops=3))
    assert eng.lib.b9_logpost(eng._ctx, None, 1, None, None) == abi.B9_ERR_INVALID


def test_reload_pack_then_same_stars():
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=200, wd_frac=0.1)
    eng = engine.Engine(pack, stars, priors, options)
    a = eng.logpost(cl["truth"][None, :])[0]
    pack2_d = synth.make_pack("parsec", 8, n_feh=4, n_age=8, n_eep=90, ifmr_id=abi.IFMR_SALARIS_LIN)
    pack2_d["m_wd_up"] = 7.0                                   # changes the IMF normalisation folded into the star constants
    pack2 = abi.make_pack(pack2_d)
    eng.load_pack(pack2)                                       # stars must be re-derived against the new pack
    b = eng.logpost(cl["truth"][None, :])[0]
    want = oracle.Oracle(pack2, stars, priors, options).logpost(cl["truth"][None, :])[0]
    assert abs(b - want) <= 1e-9 * abs(want) and a != b


def test_reload_pack_with_more_filters_same_eep_count():
    """A context that has already evaluated with a 3-filter pack (rows of 4 + 1 doubles) gets an 8-filter pack of
    the SAME isochrone length (rows of 8 + 1): the per-walker isochrone buffers must be re-sized, not re-used."""
    from base_amd import engine
    p3, cl3, pack3, stars3, priors3, options = build_problem("parsec", 3, n_stars=300)
    eng = engine.Engine(pack3, stars3, priors3, options)
    par3 = synth.walker_params(cl3["truth"], 6)
    a = eng.logpost(par3)
    assert _err(a, oracle.Oracle(pack3, stars3, priors3, options).logpost(par3)) <= 1e-9
    p8, cl8, pack8, stars8, priors8, _ = build_problem("parsec", 8, n_stars=300)
    assert p8["iso_n_eep"].max() == p3["iso_n_eep"].max()
    eng.load_pack(pack8)
    eng.load_stars(stars8)
    eng.set_priors(priors8)
    par8 = synth.walker_params(cl8["truth"], 6)
    got, want = eng.logpost(par8, perstar=True), oracle.Oracle(pack8, stars8, priors8, options).logpost(par8, perstar=True)
    assert _err(got[1], want[1]) <= 1e-9 and _err(got[0], want[0]) <= 1e-9
    # and a sampler block on the re-sized buffers reproduces a fresh context's chain
    free = np.array([abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS], dtype=np.int32)
    chol = np.diag([2e-4, 1e-3, 5e-4, 3e-4])
    ids = np.arange(6, dtype=np.int32)
    r1 = eng.mcmc_run_block(par8, got[0], ids, free, chol, 5, 0, 12)
    fresh = engine.Engine(pack8, stars8, priors8, options)
    r2 = fresh.mcmc_run_block(par8, fresh.logpost(par8), ids, free, chol, 5, 0, 12)
    np.testing.assert_array_equal(r1[2], r2[2])
    np.testing.assert_array_equal(r1[3], r2[3])


def test_load_stars_mass_validation():
    """NaN / infinite mass1 and mass ratios outside [0, 1] are input errors; mass1 <= 0 is an error in given-mass mode
    only (the marginalised mode takes mass1 as a hint)."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=40)
    for key, idx, val in (("mass1", 3, np.nan), ("mass1", 4, np.inf), ("mass_ratio", 5, -0.1), ("mass_ratio", 6, 1.5), ("mass_ratio", 7, np.nan)):
        bad = dict(cl); v = np.array(cl[key]).copy(); v[idx] = val; bad[key] = v
        with pytest.raises(engine.B9Error) as e:
            engine.Engine(pack, abi.make_stars(bad), priors, options)
        assert e.value.code == abi.B9_ERR_INVALID
    zero = dict(cl); v = np.array(cl["mass1"]).copy(); v[9] = 0.0; zero["mass1"] = v
    zs = abi.make_stars(zero)
    eng = engine.Engine(pack, zs, priors, options)                      # loading is fine ...
    with pytest.raises(engine.B9Error) as e:
        eng.logpost(cl["truth"][None, :])                                # ... evaluating in given-mass mode is not
    assert e.value.code == abi.B9_ERR_INVALID
    marg = abi.make_options(abi.MODE_MARGINALISED, 1, 2, 2)
    eng.set_options(marg)
    got = eng.logpost(cl["truth"][None, :], perstar=True)
    want = oracle.Oracle(pack, zs, priors, marg).logpost(cl["truth"][None, :], perstar=True)
    assert np.isfinite(got[0][0]) and _err(got[1], want[1]) <= 1e-9 and _err(got[0], want[0]) <= 1e-9


def test_all_stars_heavy_and_all_unused():
    """Every star above the AGB tip (old walker), and a star with no usable filter."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=300)
    cl = dict(cl)
    sig = np.array(cl["sigma"]); sig[7, :] = -1.0; cl["sigma"] = sig
    stars = abi.make_stars(cl)
    eng = engine.Engine(pack, stars, priors, options)
    par = cl["truth"].copy(); par[abi.P_LOGAGE] = pack_d["log_age"][-1] - 1e-6     # oldest isochrone: lowest tip
    params = np.stack([par, cl["truth"]])
    got, want = eng.logpost(params, perstar=True), oracle.Oracle(pack, stars, priors, options).logpost(params, perstar=True)
    assert _err(got[1], want[1]) <= 1e-9 and _err(got[0], want[0]) <= 1e-9
    heavy = (cl["mass1"] > 0.9).sum()
    assert heavy > 20       # the old walker really sends stars down the WD branch


def test_mcmc_block_argument_errors_and_zero_steps():
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("dsed", 8, n_stars=100)
    eng = engine.Engine(pack, stars, priors, options)
    start = synth.walker_params(cl["truth"], 2)
    lp = eng.logpost(start)
    out = eng.mcmc_run_block(start, lp, np.arange(2), np.array([0, 2]), np.eye(2) * 1e-3, 1, 0, 0)
    np.testing.assert_array_equal(out[0], start)
    assert out[4] == 0
    with pytest.raises(engine.B9Error):
        eng.mcmc_run_block(start, lp, np.arange(2), np.array([0, 99]), np.eye(2), 1, 0, 5)
    one = eng.mcmc_run_block(start, lp, np.arange(2), np.array([0, 2]), np.eye(2) * 1e-3, 1, 0, 1)
    two = eng.mcmc_run_block(start, lp, np.arange(2), np.array([0, 2]), np.eye(2) * 1e-3, 1, 0, 2)
    np.testing.assert_array_equal(one[2][0], two[2][0])        # the first step does not depend on the block length


@pytest.mark.parametrize("name,n_filt,n_y,n_pops", [("girardi", 3, 1, 1), ("dsed", 5, 3, 2)])
def test_parameters_exactly_on_grid_nodes_and_edges(name, n_filt, n_y, n_pops):
    """Bracket conventions bite when a cluster parameter equals a grid node (first, interior, last) or a
    star's mass equals an isochrone node: the HIP path must agree with the oracle there too, including
    on which side of the support (-inf) each edge falls."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem(name, n_filt, n_stars=200, wd_frac=0.1, n_y=n_y, n_pops=n_pops, seed=8)
    la, fe, yy = pack_d["log_age"], pack_d["feh"], pack_d["y"]
    rows = []
    for a in (la[0], la[1], la[len(la) // 2], la[-2], la[-1], np.nextafter(la[-1], 0), np.nextafter(la[0], 99),
              np.nextafter(la[-1], 99), np.nextafter(la[0], 0)):
        for f in (fe[0], fe[1], fe[-1], np.nextafter(fe[-1], -99), np.nextafter(fe[0], 99), np.nextafter(fe[-1], 99), np.nextafter(fe[0], -99)):
            for y in (yy[0], yy[-1], yy[len(yy) // 2]):
                p = cl["truth"].copy()
                p[abi.P_LOGAGE], p[abi.P_FEH], p[abi.P_Y], p[abi.P_Y2] = a, f, y, yy[-1 if n_pops == 2 else 0]
                rows.append(p)
    params = np.array(rows)
    pr = abi.make_priors(log_age_min=la[0] - 1.0, log_age_max=la[-1] + 1.0)        # let the GRID decide the support
    eng = engine.Engine(pack, stars, pr, options)
    orc = oracle.Oracle(pack, stars, pr, options)
    got, want = eng.logpost(params, perstar=True), orc.logpost(params, perstar=True)
    assert np.array_equal(np.isfinite(got[0]), np.isfinite(want[0]))
    assert np.isfinite(want[0]).sum() >= len(rows) // 3 and (~np.isfinite(want[0])).sum() >= 1
    assert _err(got[1], want[1]) <= 1e-9 and _err(got[0], want[0]) <= 1e-9
    # masses exactly on isochrone nodes (and on its two ends)
    p = cl["truth"].copy()
    first, mass, _, tip = orc.derive_isochrone(p)
    cl2 = dict(cl)
    m = np.asarray(cl["mass1"]).copy()
    k = min(len(m), len(mass))
    m[:k] = mass[np.linspace(0, len(mass) - 1, k).astype(int)]
    cl2["mass1"] = m
    st2 = abi.make_stars(cl2)
    eng2 = engine.Engine(pack, st2, priors, options)
    got = eng2.logpost(p[None, :], perstar=True)
    want = oracle.Oracle(pack, st2, priors, options).logpost(p[None, :], perstar=True)
    assert _err(got[1], want[1]) <= 1e-9 and _err(got[0], want[0]) <= 1e-9


_DEVICE_PTR_SCRIPT = r"""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
torch.cuda.init()                      # torch's HIP runtime first: the process must hold ONE copy of libamdhip64
from base_amd import abi, engine, synth
from conftest import build_problem
pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=500, wd_frac=0.05)
eng = engine.Engine(pack, stars, priors, options)
params = synth.walker_params(cl["truth"], 6, seed=3)
params[5, abi.P_FEH] = pack_d["feh"][-1] + 1.0                       # one walker outside the grid
want_lp, want_ps = eng.logpost(params, perstar=True)
dev = torch.device("cuda", eng.device_id())
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    d_par = torch.from_numpy(params).to(dev)
    d_lp = torch.full((6,), 123.0, dtype=torch.float64, device=dev)
    d_ps = torch.zeros((6, 500), dtype=torch.float64, device=dev)
    eng.logpost_device(d_par.data_ptr(), 6, d_lp.data_ptr(), d_ps.data_ptr(), stream.cuda_stream)
    stream.synchronize()
assert np.array_equal(d_lp.cpu().numpy(), want_lp) and np.array_equal(d_ps.cpu().numpy(), want_ps)
assert np.isneginf(want_lp[5]) and np.isfinite(want_lp[:5]).all()
print("device-pointer path ok")
"""


def test_logpost_device_with_caller_buffers_and_stream():
    """b9_logpost_device: parameters and outputs stay in HBM (caller's torch tensors) and the launches go on the
    caller's stream -- the zero-copy form a multi-GPU driver hands straight to RCCL.  Runs in a child process
    because torch must initialise its own HIP runtime before libbase9hip.so is loaded (as bench.py does)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DEVICE_PTR_SCRIPT], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "device-pointer path ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("n_pops,mode", [(1, abi.MODE_GIVEN_MASS), (2, abi.MODE_GIVEN_MASS), (1, abi.MODE_MARGINALISED)])
def test_non_finite_parameters_are_outside_the_support(n_pops, mode):
    """NaN / +-inf / +-1e300 in any parameter: no hang, and the same verdict as the oracle (finite or -inf, never NaN)."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, _ = build_problem("dsed", 5, n_stars=300, wd_frac=0.1, n_y=3 if n_pops == 2 else 1, n_pops=n_pops, seed=4)
    opt = abi.make_options(mode=mode, n_pops=n_pops, marg_iso_increm=2, marg_n_q=2)
    eng, orc = engine.Engine(pack, stars, priors, opt), oracle.Oracle(pack, stars, priors, opt)
    rows = []
    for k in range(abi.B9_NPARAM):
        for bad in (np.nan, np.inf, -np.inf, 1e300, -1e300):
            r = cl["truth"].copy(); r[k] = bad; rows.append(r)
    rows = np.array(rows)
    got = np.concatenate([eng.logpost(rows[i:i + 20]) for i in range(0, len(rows), 20)])
    want = orc.logpost(rows)
    assert not np.isnan(got).any() and not np.isnan(want).any()
    assert np.array_equal(np.isfinite(got), np.isfinite(want)) and np.isfinite(want).sum() >= 20
    assert _err(got, want) <= 1e-9


def test_load_stars_rejects_non_finite_photometry():
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 4, n_stars=20)
    for mutate, text in ((lambda o, s: o.__setitem__(5, np.nan), "finite observation"), (lambda o, s: o.__setitem__(6, np.inf), "finite observation"),
                         (lambda o, s: s.__setitem__(7, np.nan), "sigma is NaN"), (lambda o, s: s.__setitem__(8, 1e-300), "1e-150"),
                         (lambda o, s: s.__setitem__(9, np.inf), "sigma < inf")):
        c2 = dict(cl); o, s = np.array(cl["obs"], copy=True), np.array(cl["sigma"], copy=True)
        mutate(o, s); c2["obs"], c2["sigma"] = o, s
        with pytest.raises(RuntimeError) as e:
            engine.Engine(pack, abi.make_stars(c2), priors, options)
        assert text in str(e.value)
    c2 = dict(cl); o, s = np.array(cl["obs"], copy=True), np.array(cl["sigma"], copy=True)
    o[3] = np.nan; s[3] = -1.0                                              # an UNUSED filter may hold anything
    c2["obs"], c2["sigma"] = o, s
    st2 = abi.make_stars(c2)
    got = engine.Engine(pack, st2, priors, options).logpost(cl["truth"][None, :])
    assert np.isfinite(got[0]) and _err(got, oracle.Oracle(pack, st2, priors, options).logpost(cl["truth"][None, :])) <= 1e-9


@pytest.mark.parametrize("case", ["big_grid", "no_wd_tables", "big_grid_two_pops"])
def test_heavy_role_staging_variants(case):
    """The heavy-star role's LDS staging has two layouts the BASELINE shapes never reach: a (FeH, Y, age) grid too large
    for the whole AGB-tip table in LDS (> 2048 isochrones: only the candidates' corner columns are staged, after the
    headers are known), and a pack without WD tables (the packed axes are the age axis alone; stars above the tip have
    no flux).  Log-posterior and the fused sampler's chain against the oracle / the host twin."""
    from base_amd import engine, mcmc
    if case == "no_wd_tables":
        pack_d = synth.make_pack("parsec", 8, wd=False, n_feh=4, n_age=8, n_eep=90)
        n_pops, wd_frac = 1, 0.0
    else:
        n_pops = 2 if case.endswith("two_pops") else 1
        pack_d = synth.make_pack("parsec", 8, n_y=3, n_feh=13, n_age=60, n_eep=40)      # 13 x 3 x 60 = 2340 isochrones
        wd_frac = 0.06
    truth = synth.default_params(pack_d)
    cl = synth.make_cluster(pack_d, 700, seed=31, truth=truth, wd_frac=wd_frac, n_pops=n_pops)
    pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
    priors, options = synth.default_priors(pack_d, truth, n_pops), abi.make_options(abi.MODE_GIVEN_MASS, n_pops, 4, 4)
    eng = engine.Engine(pack, stars, priors, options)
    params = synth.walker_params(truth, 4, seed=5, n_pops=n_pops, scale=0.2)
    got = eng.logpost(params, perstar=True)
    want = oracle.Oracle(pack, stars, priors, options).logpost(params, perstar=True)
    assert _err(got[1], want[1]) <= 1e-9 and _err(got[0], want[0]) <= 1e-9
    if case != "no_wd_tables":
        tips = [eng.derive_isochrone(params[0])[3]]
        assert (cl["mass1"] > min(tips)).sum() >= 10                              # the role had stars to evaluate
    free = np.array(mcmc.DEFAULT_FREE if n_pops == 1 else mcmc.DEFAULT_FREE + (abi.P_Y, abi.P_Y2, abi.P_LAMBDA))
    chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
    start = synth.walker_params(truth, 3, seed=6, n_pops=n_pops, scale=0.05)
    lp0 = eng.logpost(start)
    ids = np.arange(3)
    host = mcmc.HostBlockRunner(eng.logpost).run(start, lp0, ids, free, chol, 9, 1000, 12)
    dev = mcmc.DeviceBlockRunner(eng).run(start, lp0, ids, free, chol, 9, 1000, 12)
    assert dev[4] == host[4]
    np.testing.assert_allclose(dev[2], host[2], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(dev[3], host[3], rtol=1e-10)


def test_shader_clock_between_two_stamps():
    """b9_clock_stamp / b9_clock_mhz (ABI 5): the shader clock observed in-kernel over a stretch of the context's stream --
    delta(s_memtime) / delta(s_memrealtime) x 100 MHz, median over the compute units -- is a plausible MI355X clock; asking before
    any stamp is a state error."""
    from base_amd import engine
    pack_d, cl, pack, stars, priors, options = build_problem("parsec", 8, n_stars=20000, small=False)
    eng = engine.Engine(pack, stars, priors, options)
    with pytest.raises(engine.B9Error) as e:
        eng.clock_mhz()
    assert e.value.code == abi.B9_ERR_STATE
    params = synth.walker_params(cl["truth"], 8)
    eng.logpost(params)
    eng.clock_stamp(0)
    for _ in range(200):
        eng.logpost(params)
    eng.clock_stamp(1)
    c = eng.clock_mhz()
    assert 300.0 < c["mhz_min_cu"] <= c["mhz"] <= c["mhz_max_cu"] < 2600.0, c
    assert c["mhz_max_cu"] - c["mhz_min_cu"] < 0.1 * c["mhz"], c          # per-CU differences: no counter offsets in them
    assert c["ref_seconds"] > 1e-3
    with pytest.raises(engine.B9Error):
        eng.clock_stamp(2)
