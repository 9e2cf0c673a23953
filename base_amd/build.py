"""Builds the native pieces in-tree with hipcc / g++ (no cmake, no JIT cache).

  base_amd/csrc/libbase9hip.so   the HIP kernels + C ABI (gfx950 only)
  base_amd/host/...              the C++ host programs (added by build_host)

hipcc cross-compiles for gfx950 without a GPU, so this runs in the dev container; the built
.so travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.path.join(CSRC, "libbase9hip.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _newer(target: str, sources: List[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd: List[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("build failed: " + " ".join(cmd))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


HIP_SOURCES = ("b9_kernels.hip", "b9_capi_ctx.cpp", "b9_capi_stage.cpp", "b9_capi_plan.cpp", "b9_capi_margplan.cpp", "b9_capi_eval.cpp", "b9_capi_blocks.cpp")


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """One object per source under build/obj/ (recompiled when it or any header is newer), compiled in parallel, then linked."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "base9_hip.h")]
    if not force and _newer(HIP_LIB, srcs + hdrs):
        return HIP_LIB
    obj_dir = os.path.join(ROOT, "build", "obj")
    os.makedirs(obj_dir, exist_ok=True)
    objs = [os.path.join(obj_dir, os.path.basename(s) + ".o") for s in srcs]

    def compile_one(so):
        src, obj = so
        if not force and _newer(obj, [src] + hdrs):
            return
        cmd = [HIPCC] + HIP_FLAGS + ["-c", "-o", obj, "-x", "hip", src]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        _run(cmd)
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        list(ex.map(compile_one, zip(srcs, objs)))
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs)
    return HIP_LIB


def source_hash() -> str:
    """sha256 over the kernel sources and the ABI header, in name order: which build a measured profile belongs to
    (tools/profile_summary.py stores it; bench.py drops the profile's counters when the tree it runs in has another)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    for f in files + [os.path.join(ROOT, "include", "base9_hip.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build_all(force: bool = False) -> None:
    build_hip(force)
    try:
        from . import host_build  # noqa: WPS433  (optional until the host programs exist)
    except ImportError:
        return
    host_build.build_host(force)


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(HIP_LIB)
