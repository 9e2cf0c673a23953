"""ctypes mirror of include/base9_hip.h (the C ABI of the hot path).

Plumbing only: struct layouts, argtypes, and helpers that pin numpy arrays behind the plain
pointers the ABI takes.  The same structs are accepted by the CPU oracle (tests only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

B9_NPARAM = 12
(P_LOGAGE, P_Y, P_FEH, P_MOD, P_ABS, P_CARBONICITY, P_IFMR_INTERCEPT, P_IFMR_SLOPE,
 P_IFMR_QUAD, P_Y2, P_LAMBDA, P_RESERVED) = range(12)
PARAM_NAMES = ["logAge", "Y", "FeH", "modulus", "absorption", "carbonicity",
               "IFMRconst", "IFMRlin", "IFMRquad", "Y2", "lambda", "reserved"]

BLOCK_CONTINUE, BLOCK_ASYNC, BLOCK_ROWS_EVENT = 1, 2, 4
STAGE_MSRG, STAGE_WD, STAGE_NSBH, STAGE_BD, STAGE_DNE = 1, 3, 4, 5, 9
IFMR_WEIDEMANN, IFMR_WILLIAMS, IFMR_SALARIS_LIN, IFMR_SALARIS_PW, IFMR_LINEAR, IFMR_QUADRATIC = range(6)
MODE_GIVEN_MASS, MODE_MARGINALISED = 0, 1
MAG_NOFLUX = 99.999

B9_OK, B9_ERR_NO_DEVICE, B9_ERR_INVALID, B9_ERR_STATE, B9_ERR_HIP, B9_ERR_CAPACITY = 0, -1, -2, -3, -4, -5

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class b9_pack(C.Structure):
    _fields_ = [
        ("n_filt", C.c_int32),
        ("n_feh", C.c_int32), ("n_y", C.c_int32), ("n_age", C.c_int32),
        ("feh", _dp), ("y", _dp), ("log_age", _dp),
        ("iso_first_eep", _ip), ("iso_n_eep", _ip), ("iso_offset", _lp),
        ("n_points", C.c_int64),
        ("mass", _dp), ("mags", _dp), ("abs_coeff", _dp),
        ("n_wc_carb", C.c_int32), ("n_wc_mass", C.c_int32),
        ("wc_carb", _dp), ("wc_mass", _dp), ("wc_n_age", _ip), ("wc_offset", _lp), ("n_wc_points", C.c_int64),
        ("wc_log_age", _dp), ("wc_log_teff", _dp), ("wc_log_radius", _dp),
        ("n_at_type", C.c_int32), ("n_at_logg", C.c_int32), ("n_at_teff", C.c_int32),
        ("at_logg", _dp), ("at_log_teff", _dp), ("at_mags", _dp),
        ("ifmr_id", C.c_int32), ("reserved0", C.c_int32),
        ("m_wd_up", C.c_double),
    ]


class b9_stars(C.Structure):
    _fields_ = [
        ("n_stars", C.c_int32), ("n_filt", C.c_int32),
        ("obs", _dp), ("sigma", _dp), ("mass1", _dp), ("mass_ratio", _dp), ("clust_prior", _dp),
        ("stage", _ip), ("wd_type", _ip),
        ("filter_prior_min", _dp), ("filter_prior_max", _dp),
    ]


class b9_priors(C.Structure):
    _fields_ = [("mean", C.c_double * B9_NPARAM), ("var", C.c_double * B9_NPARAM),
                ("log_age_min", C.c_double), ("log_age_max", C.c_double)]


class b9_options(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_pops", C.c_int32),
                ("marg_iso_increm", C.c_int32), ("marg_n_q", C.c_int32)]


class b9_tuning(C.Structure):
    """Launch-plan tuning (include/base9_hip.h); all zeros = automatic."""
    _fields_ = [(n, C.c_int32) for n in ("tiles_per_block", "derive_parts", "derive_order", "heavy_parts", "two_launch_steps",
                                         "marg_no_pruning", "timing_group", "plan_debug", "tree_depth", "marg_piece_units")] + \
               [("reserved", C.c_int32 * 6)]


class b9_mcmc_block(C.Structure):
    _fields_ = [("n_walkers", C.c_int32), ("n_free", C.c_int32),
                ("free_idx", _ip), ("chol", _dp), ("walker_ids", _ip),
                ("seed", C.c_uint64), ("step0", C.c_int64),
                ("n_steps", C.c_int32), ("flags", C.c_int32),
                ("params", _dp), ("logpost", _dp), ("samples", _dp), ("lps", _dp),
                ("n_accept", C.c_int64),
                ("row_origin", _dp), ("rows", _dp), ("d_rows", C.c_void_p), ("rows_ready", C.c_void_p)]


def row_doubles(d: int) -> int:
    """B9_ROW_DOUBLES(d): length of a walker's block summary row."""
    return 15 + d + d * d


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


class Pinned:
    """A ctypes struct plus the numpy arrays its pointers refer to (kept alive together)."""

    def __init__(self, struct, keep: Dict[str, np.ndarray]):
        self.struct = struct
        self.keep = keep

    def byref(self):
        return C.byref(self.struct)


def make_pack(d: Dict) -> Pinned:
    """Build a b9_pack from a dict of arrays (see synth.make_pack for the keys)."""
    k: Dict[str, np.ndarray] = {}
    for name in ("feh", "y", "log_age", "mass", "mags", "abs_coeff", "wc_carb", "wc_mass",
                 "wc_log_age", "wc_log_teff", "wc_log_radius", "at_logg", "at_log_teff", "at_mags"):
        k[name] = _f64(d.get(name, np.zeros(0)))
    # WD cooling tracks: ragged (wc_n_age / wc_offset given) or the older rectangular form (one shared age axis, tables
    # [carb][mass][age]), which is expanded to one copy of the axis per track
    n_tracks = max(1, len(k["wc_carb"])) * len(k["wc_mass"])
    if "wc_n_age" in d:
        k["wc_n_age"] = np.ascontiguousarray(d["wc_n_age"], dtype=np.int32).ravel()
        k["wc_offset"] = np.ascontiguousarray(d["wc_offset"], dtype=np.int64).ravel()
    else:
        n_age = len(k["wc_log_age"])
        k["wc_n_age"] = np.full(n_tracks, n_age, dtype=np.int32)
        k["wc_offset"] = np.arange(n_tracks, dtype=np.int64) * n_age
        k["wc_log_age"] = np.ascontiguousarray(np.tile(k["wc_log_age"], n_tracks))
    assert k["wc_n_age"].size == n_tracks and k["wc_log_teff"].size == k["wc_log_age"].size == k["wc_log_radius"].size
    k["iso_first_eep"] = np.ascontiguousarray(d["iso_first_eep"], dtype=np.int32).ravel()
    k["iso_n_eep"] = np.ascontiguousarray(d["iso_n_eep"], dtype=np.int32).ravel()
    k["iso_offset"] = np.ascontiguousarray(d["iso_offset"], dtype=np.int64).ravel()
    p = b9_pack()
    p.n_filt = int(d["n_filt"])
    p.n_feh, p.n_y, p.n_age = len(k["feh"]), len(k["y"]), len(k["log_age"])
    assert k["iso_n_eep"].size == p.n_feh * p.n_y * p.n_age
    p.n_points = int(k["mass"].size)
    assert k["mags"].size == p.n_points * p.n_filt
    for name in ("feh", "y", "log_age", "mass", "mags", "abs_coeff", "wc_carb", "wc_mass",
                 "wc_log_age", "wc_log_teff", "wc_log_radius", "at_logg", "at_log_teff", "at_mags"):
        setattr(p, name, _ptr(k[name], _dp))
    p.iso_first_eep = _ptr(k["iso_first_eep"], _ip)
    p.iso_n_eep = _ptr(k["iso_n_eep"], _ip)
    p.iso_offset = _ptr(k["iso_offset"], _lp)
    p.n_wc_carb, p.n_wc_mass, p.n_wc_points = len(k["wc_carb"]), len(k["wc_mass"]), int(k["wc_log_age"].size)
    p.wc_n_age = _ptr(k["wc_n_age"], _ip)
    p.wc_offset = _ptr(k["wc_offset"], _lp)
    p.n_at_logg, p.n_at_teff = len(k["at_logg"]), len(k["at_log_teff"])
    p.n_at_type = int(d.get("n_at_type", 0 if p.n_at_teff == 0 else k["at_mags"].size // max(1, p.n_at_logg * p.n_at_teff * p.n_filt)))
    p.ifmr_id = int(d.get("ifmr_id", IFMR_WILLIAMS))
    p.m_wd_up = float(d.get("m_wd_up", 8.0))
    return Pinned(p, k)


def make_stars(d: Dict) -> Pinned:
    k: Dict[str, np.ndarray] = {}
    n = int(len(d["mass1"]))
    nf = int(d["n_filt"])
    for name in ("obs", "sigma", "mass1", "mass_ratio", "clust_prior", "filter_prior_min", "filter_prior_max"):
        k[name] = _f64(d[name]).ravel()
    assert k["obs"].size == n * nf and k["sigma"].size == n * nf
    k["stage"] = np.ascontiguousarray(d.get("stage", np.full(n, STAGE_MSRG)), dtype=np.int32)
    k["wd_type"] = np.ascontiguousarray(d.get("wd_type", np.zeros(n)), dtype=np.int32)
    s = b9_stars()
    s.n_stars, s.n_filt = n, nf
    for name in ("obs", "sigma", "mass1", "mass_ratio", "clust_prior", "filter_prior_min", "filter_prior_max"):
        setattr(s, name, _ptr(k[name], _dp))
    s.stage = _ptr(k["stage"], _ip)
    s.wd_type = _ptr(k["wd_type"], _ip)
    return Pinned(s, k)


def make_priors(mean=None, var=None, log_age_min=-np.inf, log_age_max=np.inf) -> b9_priors:
    pr = b9_priors()
    mean = np.zeros(B9_NPARAM) if mean is None else np.asarray(mean, dtype=np.float64)
    var = np.zeros(B9_NPARAM) if var is None else np.asarray(var, dtype=np.float64)
    for i in range(B9_NPARAM):
        pr.mean[i] = float(mean[i])
        pr.var[i] = float(var[i])
    pr.log_age_min, pr.log_age_max = float(log_age_min), float(log_age_max)
    return pr


def make_options(mode=MODE_GIVEN_MASS, n_pops=1, marg_iso_increm=8, marg_n_q=8) -> b9_options:
    return b9_options(int(mode), int(n_pops), int(marg_iso_increm), int(marg_n_q))


REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB_PATH = os.path.join(REPO_ROOT, "base_amd", "csrc", "libbase9hip.so")

#: every symbol include/base9_hip.h declares (checked by tests/test_abi.py against the header)
ABI_SYMBOLS = [
    "b9_abi_version", "b9_ctx_create", "b9_ctx_destroy", "b9_last_error",
    "b9_load_pack", "b9_load_stars", "b9_set_priors", "b9_set_options", "b9_set_tuning", "b9_get_tuning",
    "b9_logpost", "b9_logpost_device", "b9_mcmc_run_block", "b9_mcmc_wait", "b9_sample_mass", "b9_derive_isochrone",
    "b9_max_eep", "b9_device_id", "b9_bytes_per_star_eval", "b9_step_tiles_per_block", "b9_step_depth",
    "b9_enable_timing", "b9_kernel_time_ms", "b9_calibrate_timing", "b9_clock_stamp", "b9_clock_mhz",
]


def load_hip_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libbase9hip.so and declare its prototypes.  Raises if it is not built."""
    path = path or os.environ.get("B9_HIP_LIB") or HIP_LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the hot path)")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.b9_abi_version.restype = C.c_int
    lib.b9_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.b9_ctx_destroy.argtypes = [vp]
    lib.b9_ctx_destroy.restype = None
    lib.b9_last_error.argtypes = [vp]
    lib.b9_last_error.restype = C.c_char_p
    lib.b9_load_pack.argtypes = [vp, C.POINTER(b9_pack)]
    lib.b9_load_stars.argtypes = [vp, C.POINTER(b9_stars)]
    lib.b9_set_priors.argtypes = [vp, C.POINTER(b9_priors)]
    lib.b9_set_options.argtypes = [vp, C.POINTER(b9_options)]
    lib.b9_set_tuning.argtypes = [vp, C.POINTER(b9_tuning)]
    lib.b9_get_tuning.argtypes = [vp, C.POINTER(b9_tuning)]
    lib.b9_logpost.argtypes = [vp, _dp, C.c_int32, _dp, _dp]
    lib.b9_logpost_device.argtypes = [vp, vp, C.c_int32, vp, vp, vp]
    lib.b9_mcmc_run_block.argtypes = [vp, C.POINTER(b9_mcmc_block)]
    lib.b9_mcmc_wait.argtypes = [vp, C.POINTER(b9_mcmc_block)]
    lib.b9_sample_mass.argtypes = [vp, _dp, C.c_int32, C.c_uint64, C.c_int64, _dp, _dp, _dp, _ip]
    lib.b9_derive_isochrone.argtypes = [vp, _dp, C.c_int32, C.c_int32, _dp, _dp, _ip, _ip, _dp]
    lib.b9_max_eep.argtypes = [vp]
    lib.b9_device_id.argtypes = [vp]
    lib.b9_bytes_per_star_eval.argtypes = [vp]
    lib.b9_step_tiles_per_block.argtypes = [vp, C.c_int32]
    lib.b9_step_depth.argtypes = [vp, C.c_int32]
    lib.b9_enable_timing.argtypes = [vp, C.c_int]
    lib.b9_kernel_time_ms.argtypes = [vp, C.c_int, _dp, _ip]
    lib.b9_calibrate_timing.argtypes = [vp, _dp]
    lib.b9_clock_stamp.argtypes = [vp, C.c_int32]
    lib.b9_clock_mhz.argtypes = [vp, _dp, _dp, _dp, _dp]
    return lib
