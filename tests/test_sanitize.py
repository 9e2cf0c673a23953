"""CPU hygiene: the C oracle and the C++ host parsers built and run under
AddressSanitizer + UndefinedBehaviorSanitizer.  (GPU sanitizers are not available on this pool.)"""
import os
import subprocess

import numpy as np

from base_amd import abi, synth
from conftest import build_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_parsers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_driver")
    host = os.path.join(ROOT, "base_amd", "host")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "tests", "sanitize", "driver.cpp"),
           os.path.join(host, "b9host.cpp"), "-x", "c", os.path.join(ROOT, "oracle", "b9_oracle.c"), "-lm"]
    # b9host.cpp references run_mcmc -> the C ABI; the driver never calls it, so stub the symbols it needs
    stub = tmp_path / "stub.c"
    stub.write_text('#include "base9_hip.h"\n'
                    "int b9_logpost(b9_ctx *c, const double *p, int32_t n, double *o, double *s) { (void)c; (void)p; (void)n; (void)o; (void)s; return B9_ERR_NO_DEVICE; }\n"
                    "int b9_mcmc_run_block(b9_ctx *c, b9_mcmc_block *b) { (void)c; (void)b; return B9_ERR_NO_DEVICE; }\n"
                    "int b9_mcmc_wait(b9_ctx *c, b9_mcmc_block *b) { (void)c; (void)b; return B9_ERR_NO_DEVICE; }\n"
                    'const char *b9_last_error(const b9_ctx *c) { (void)c; return "stub"; }\n')
    cmd += [str(stub), "-I", os.path.join(ROOT, "include")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    pack_d, cl, *_ = build_problem("parsec", 8, n_stars=150, wd_frac=0.1)
    root = synth.write_models_dir(pack_d, str(tmp_path / "models"))
    phot = synth.write_phot(cl, pack_d["filters"], str(tmp_path / "c.phot"))
    yml = synth.write_yaml(str(tmp_path / "base9.yaml"), phot, root, str(tmp_path / "o"), cl["truth"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, root, "parsec", phot, yml], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert r.stdout.startswith("OK 150") and "ERROR" not in r.stderr and "runtime error" not in r.stderr
    # the sanitized oracle agrees with the regular build
    import oracle
    want = oracle.Oracle(abi.make_pack(pack_d), abi.make_stars(dict(cl, filter_prior_min=np.where(np.asarray(cl["sigma"]) > 0, cl["obs"], np.inf).min(axis=0),
                                                                     filter_prior_max=np.where(np.asarray(cl["sigma"]) > 0, cl["obs"], -np.inf).max(axis=0))),
                         abi.make_priors(log_age_min=pack_d["log_age"][0], log_age_max=pack_d["log_age"][-1]), abi.make_options())
    row = np.zeros(abi.B9_NPARAM)
    t = cl["truth"]
    for k in (abi.P_LOGAGE, abi.P_FEH, abi.P_Y, abi.P_MOD, abi.P_ABS):
        row[k] = t[k]
    row[abi.P_CARBONICITY] = 0.38
    got = float(r.stdout.split()[2])
    assert abs(got - want.logpost(row[None, :])[0]) < 1e-5      # printed with 6 decimals
