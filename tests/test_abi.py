"""CPU tests of the C-ABI boundary: the libraries load, export exactly what include/base9_hip.h (the hot path) and
include/base9_host.h (the C++ host driver's C surface) declare, mirror the struct layouts, and refuse to run without
a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from base_amd import abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "base9_hip.h")
HOST_HEADER = os.path.join(ROOT, "include", "base9_host.h")


@pytest.fixture(scope="module")
def lib():
    build.build_hip()
    return abi.load_hip_library()


def _declared_functions(header=HEADER, prefix="b9_"):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef[^;]*;", "", src, flags=re.S)          # function-pointer typedefs are not exports
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_match_binding_table():
    assert _declared_functions() == sorted(abi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} declared in base9_hip.h but not exported"
    assert lib.b9_abi_version() == 6


def test_host_library_exports_every_declared_symbol():
    from base_amd import host_build, hostlib
    host_build.build_host()
    assert _declared_functions(HOST_HEADER, "b9h_") == sorted(hostlib.HOST_SYMBOLS)
    hl = hostlib.load()
    for name in hostlib.HOST_SYMBOLS:
        assert hasattr(hl, name), f"{name} declared in base9_host.h but not exported"
    # the exchange of a one-rank run needs no GPU; an RCCL exchange fails loudly without one
    ex = hostlib.Exchange.local()
    assert ex.world == 1 and ex.max(3.5) == 3.5
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(hostlib.HostError):
            hostlib.Exchange.rccl(0, 1, 0)


def test_struct_layout_matches_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "base9_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(b9_pack), sizeof(b9_stars), sizeof(b9_priors), sizeof(b9_options));
  printf("%zu %zu %zu %d\n", sizeof(b9_mcmc_block), offsetof(b9_mcmc_block, row_origin), offsetof(b9_mcmc_block, rows_ready), B9_ROW_DOUBLES(4));
  printf("%zu %zu %zu %zu\n", offsetof(b9_pack, mass), offsetof(b9_pack, at_mags), offsetof(b9_pack, m_wd_up), offsetof(b9_stars, filter_prior_max));
  printf("%d %zu %zu\n", B9_NPARAM, sizeof(b9_tuning), offsetof(b9_tuning, plan_debug));
  return 0; }''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    got = [int(x) for x in out]
    assert got[:4] == [C.sizeof(abi.b9_pack), C.sizeof(abi.b9_stars), C.sizeof(abi.b9_priors), C.sizeof(abi.b9_options)]
    assert got[4:8] == [C.sizeof(abi.b9_mcmc_block), abi.b9_mcmc_block.row_origin.offset, abi.b9_mcmc_block.rows_ready.offset, abi.row_doubles(4)]
    got = got[:4] + got[8:]
    assert got[4:8] == [abi.b9_pack.mass.offset, abi.b9_pack.at_mags.offset, abi.b9_pack.m_wd_up.offset,
                        abi.b9_stars.filter_prior_max.offset]
    assert got[8:] == [abi.B9_NPARAM, C.sizeof(abi.b9_tuning), abi.b9_tuning.plan_debug.offset]


def test_no_cpu_fallback(lib):
    """Without a GPU the product path must fail loudly, not fall back to anything."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = lib.b9_ctx_create(-1, C.byref(ctx))
    assert rc == abi.B9_ERR_NO_DEVICE and not ctx.value
    assert b"no CPU fallback" in lib.b9_last_error(None)
    from base_amd import engine
    with pytest.raises(engine.B9Error):
        engine.Engine()


def test_product_does_not_reference_oracle():
    """Nothing under base_amd/ or include/ may import, link or mention loading the oracle."""
    bad = []
    for base in ("base_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"import oracle|from oracle|libb9oracle|b9o_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
