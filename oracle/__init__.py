"""Loader for the CPU oracle (libb9oracle.so).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see b9_oracle.c header).  Import this module only
from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in base_amd/
imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from base_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(native: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds).  Returns the .so path."""
    target = "native" if native else "all"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, "libb9oracle_native.so" if native else "libb9oracle.so")


def load(native: bool = False) -> C.CDLL:
    path = os.path.join(_HERE, "libb9oracle_native.so" if native else "libb9oracle.so")
    src = os.path.join(_HERE, "b9_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build(native)
    lib = C.CDLL(path)
    lib.b9o_logpost.argtypes = [C.POINTER(abi.b9_pack), C.POINTER(abi.b9_stars), C.POINTER(abi.b9_priors),
                                C.POINTER(abi.b9_options), _dp, C.c_int, _dp, _dp]
    lib.b9o_logpost.restype = C.c_int
    lib.b9o_derive_isochrone_flat.argtypes = [C.POINTER(abi.b9_pack), _dp, C.c_int, C.c_int, _dp, _dp, _ip, _ip, _dp]
    lib.b9o_derive_isochrone_flat.restype = C.c_int
    lib.b9o_log_mass_norm.argtypes = [C.c_double]
    lib.b9o_log_mass_norm.restype = C.c_double
    lib.b9o_log_prior_mass.argtypes = [C.c_double, C.c_double]
    lib.b9o_log_prior_mass.restype = C.c_double
    lib.b9o_log_prior_cluster.argtypes = [C.POINTER(abi.b9_priors), _dp, C.c_int]
    lib.b9o_log_prior_cluster.restype = C.c_double
    lib.b9o_sample_mass.argtypes = [C.POINTER(abi.b9_pack), C.POINTER(abi.b9_stars), C.POINTER(abi.b9_options), _dp, C.c_int,
                                    C.c_uint64, C.c_int64, _dp, _dp, _dp, _ip, _dp]
    lib.b9o_sample_mass.restype = C.c_int
    lib.b9o_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.b9o_philox4x32.restype = None
    lib.b9o_max_threads.restype = C.c_int
    lib.b9o_set_threads.argtypes = [C.c_int]
    lib.b9o_set_threads.restype = None
    return lib


class Oracle:
    """Thin numpy front-end over the C oracle."""

    def __init__(self, pack: abi.Pinned, stars: abi.Pinned, priors: abi.b9_priors,
                 options: abi.b9_options, native: bool = False):
        self.lib = load(native)
        self.pack, self.stars, self.priors, self.options = pack, stars, priors, options

    def logpost(self, params: np.ndarray, perstar: bool = False):
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, abi.B9_NPARAM)
        nw = params.shape[0]
        out = np.empty(nw)
        ps = np.empty((nw, self.stars.struct.n_stars)) if perstar else None
        rc = self.lib.b9o_logpost(self.pack.byref(), self.stars.byref(), C.byref(self.priors),
                                  C.byref(self.options), params.ctypes.data_as(_dp), nw,
                                  out.ctypes.data_as(_dp), ps.ctypes.data_as(_dp) if perstar else None)
        if rc != 0:
            raise RuntimeError(f"b9o_logpost failed: {rc}")
        return (out, ps) if perstar else out

    def sample_mass(self, params: np.ndarray, seed: int = 1, row0: int = 0):
        """Restatement of b9_sample_mass.  Returns (mass, ratio, member, pop, margin); margin = best key minus
        second-best key of each draw."""
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, abi.B9_NPARAM)
        nr, n = params.shape[0], self.stars.struct.n_stars
        mass, ratio, member, margin = (np.empty((nr, n)) for _ in range(4))
        pop = np.empty((nr, n), dtype=np.int32)
        rc = self.lib.b9o_sample_mass(self.pack.byref(), self.stars.byref(), C.byref(self.options), params.ctypes.data_as(_dp), nr,
                                      int(seed), int(row0), mass.ctypes.data_as(_dp), ratio.ctypes.data_as(_dp),
                                      member.ctypes.data_as(_dp), pop.ctypes.data_as(_ip), margin.ctypes.data_as(_dp))
        if rc != 0:
            raise RuntimeError(f"b9o_sample_mass failed: {rc}")
        return mass, ratio, member, pop, margin

    def derive_isochrone(self, param_row: np.ndarray, pop: int = 0, cap: int = 4096):
        return derive_isochrone(self.lib, self.pack, param_row, pop, cap)


def derive_isochrone(lib, pack: abi.Pinned, param_row, pop: int = 0, cap: int = 4096):
    row = np.ascontiguousarray(param_row, dtype=np.float64)
    nf = pack.struct.n_filt
    mass = np.empty(cap)
    mags = np.empty(cap * nf)
    first, n, tip = C.c_int32(0), C.c_int32(0), C.c_double(0)
    rc = lib.b9o_derive_isochrone_flat(pack.byref(), row.ctypes.data_as(_dp), pop, cap,
                                       mass.ctypes.data_as(_dp), mags.ctypes.data_as(_dp),
                                       C.byref(first), C.byref(n), C.byref(tip))
    if rc != 0:
        raise RuntimeError(f"b9o_derive_isochrone_flat failed: {rc}")
    return first.value, mass[:n.value].copy(), mags[:n.value * nf].reshape(n.value, nf).copy(), tip.value
