"""Builds the C++ host programs (base_amd/host/): libbase9host.so + singlePopMcmc, multiPopMcmc,
makeCMD, sampleMass.  Plain g++ against the C ABI; they link libbase9hip.so by rpath."""
from __future__ import annotations

import os
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
HOST = os.path.join(HERE, "host")
BIN = os.path.join(HOST, "bin")
CSRC = os.path.join(HERE, "csrc")
CXX = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra"]


def _run(cmd: List[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("build failed: " + " ".join(cmd))


def _stale(target: str, sources: List[str]) -> bool:
    return not os.path.exists(target) or any(os.path.getmtime(s) > os.path.getmtime(target) for s in sources)


ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")


def build_host(force: bool = False) -> None:
    """libbase9host.so = parsers + sampler (g++) + the RCCL exchange and the C surface (hipcc as a host compiler, for the
    HIP / RCCL headers: neither file holds device code); it links libbase9hip.so, librccl and libamdhip64."""
    os.makedirs(BIN, exist_ok=True)
    hdrs = [os.path.join(HOST, f) for f in ("b9host.hpp", "cli_common.hpp", "b9sampler.hpp", "b9dist.hpp")] + \
        [os.path.join(HERE, "..", "include", f) for f in ("base9_hip.h", "base9_host.h")]
    lib = os.path.join(HOST, "libbase9host.so")
    gxx_src = ["b9host.cpp", "b9sampler.cpp", "cli_common.cpp"]
    hip_src = ["b9dist.cpp", "capi_host.cpp"]
    link = ["-L" + CSRC, "-lbase9hip", "-Wl,-rpath," + CSRC, "-Wl,-rpath,$ORIGIN/../../csrc"]
    if force or _stale(lib, [os.path.join(HOST, f) for f in gxx_src + hip_src] + hdrs):
        objs = []
        for f in gxx_src:
            o = os.path.join(HOST, f.replace(".cpp", ".o"))
            _run(CXX + ["-ffp-contract=off", "-c", "-o", o, os.path.join(HOST, f)])
            objs.append(o)
        for f in hip_src:
            o = os.path.join(HOST, f.replace(".cpp", ".o"))
            _run([HIPCC, "-x", "c++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"),
                  "-c", "-o", o, os.path.join(HOST, f)])
            objs.append(o)
        _run(["g++", "-shared", "-o", lib] + objs + link + ["-L" + os.path.join(ROCM, "lib"), "-lrccl", "-lamdhip64",
                                                            "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    progs = {"singlePopMcmc": ("mcmc_main.cpp", ["-DB9_N_POPS=1"]), "multiPopMcmc": ("mcmc_main.cpp", ["-DB9_N_POPS=2"]),
             "makeCMD": ("makecmd_main.cpp", []), "sampleMass": ("samplemass_main.cpp", [])}
    for name, (src, defs) in progs.items():
        exe = os.path.join(BIN, name)
        srcp = os.path.join(HOST, src)
        if force or _stale(exe, [srcp, lib] + hdrs):
            _run(CXX + defs + ["-o", exe, srcp, "-L" + HOST, "-lbase9host", "-Wl,-rpath," + HOST, "-Wl,-rpath,$ORIGIN/.."] + link)


if __name__ == "__main__":
    build_host(force="--force" in sys.argv)
    print(BIN)
