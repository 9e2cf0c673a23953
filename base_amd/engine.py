"""Python front-end over the C ABI (include/base9_hip.h) -- used by tests, bench and the
walker-parallel driver.  Everything numeric happens in libbase9hip.so on the GPU; this class
only pins buffers and forwards calls.  It raises if the library or a GPU is missing: the hot
path has no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np

from . import abi

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class B9Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"base9_hip error {code}: {msg}")
        self.code = code


class Engine:
    """One context on one GPU: `Engine(pack, stars, priors, options).logpost(params)`."""

    def __init__(self, pack: Optional[abi.Pinned] = None, stars: Optional[abi.Pinned] = None,
                 priors: Optional[abi.b9_priors] = None, options: Optional[abi.b9_options] = None,
                 device: int = -1, lib: Optional[C.CDLL] = None):
        self.lib = lib or abi.load_hip_library()
        self._ctx = C.c_void_p()
        rc = self.lib.b9_ctx_create(int(device), C.byref(self._ctx))
        if rc != abi.B9_OK:
            msg = self.lib.b9_last_error(None).decode()
            self._ctx = C.c_void_p()
            raise B9Error(rc, msg)
        self.n_stars = 0
        self.n_filt = 0
        if pack is not None:
            self.load_pack(pack)
        if stars is not None:
            self.load_stars(stars)
        if priors is not None:
            self.set_priors(priors)
        self.options = abi.make_options()                # the context's defaults
        if options is not None:
            self.set_options(options)

    # -- plumbing -------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != abi.B9_OK:
            raise B9Error(rc, self.lib.b9_last_error(self._ctx).decode())

    def close(self) -> None:
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.b9_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- staging --------------------------------------------------------------------------
    def load_pack(self, pack: abi.Pinned) -> None:
        self._check(self.lib.b9_load_pack(self._ctx, pack.byref()))
        self.n_filt = pack.struct.n_filt

    def load_stars(self, stars: abi.Pinned) -> None:
        self._check(self.lib.b9_load_stars(self._ctx, stars.byref()))
        self.n_stars = stars.struct.n_stars

    def set_priors(self, priors: abi.b9_priors) -> None:
        self._check(self.lib.b9_set_priors(self._ctx, C.byref(priors)))

    def set_options(self, options: abi.b9_options) -> None:
        self._check(self.lib.b9_set_options(self._ctx, C.byref(options)))
        self.options = options

    def set_tuning(self, **fields) -> None:
        """b9_set_tuning: launch-plan fields of abi.b9_tuning by name (tiles_per_block=3, ...); no argument = automatic."""
        t = abi.b9_tuning()
        for k, v in fields.items():
            if not hasattr(t, k):
                raise AttributeError(f"b9_tuning has no field {k}")
            setattr(t, k, int(v))
        self._check(self.lib.b9_set_tuning(self._ctx, C.byref(t)))

    def update_tuning(self, **fields) -> None:
        """b9_get_tuning + b9_set_tuning: change the named fields only (the environment's overrides and earlier settings stay)."""
        t = abi.b9_tuning()
        self._check(self.lib.b9_get_tuning(self._ctx, C.byref(t)))
        for k, v in fields.items():
            if not hasattr(t, k):
                raise AttributeError(f"b9_tuning has no field {k}")
            setattr(t, k, int(v))
        self._check(self.lib.b9_set_tuning(self._ctx, C.byref(t)))

    # -- hot path -------------------------------------------------------------------------
    def logpost(self, params: np.ndarray, perstar: bool = False):
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, abi.B9_NPARAM)
        nw = params.shape[0]
        out = np.empty(nw)
        ps = np.empty((nw, self.n_stars)) if perstar else None
        self._check(self.lib.b9_logpost(self._ctx, params.ctypes.data_as(_dp), nw, out.ctypes.data_as(_dp),
                                        ps.ctypes.data_as(_dp) if perstar else None))
        return (out, ps) if perstar else out

    def logpost_device(self, d_params: int, n_walkers: int, d_logpost: int, d_perstar: int = 0,
                       stream: int = 0) -> None:
        """Asynchronous, device pointers (ints, e.g. torch.Tensor.data_ptr())."""
        self._check(self.lib.b9_logpost_device(self._ctx, C.c_void_p(d_params), int(n_walkers),
                                               C.c_void_p(d_logpost), C.c_void_p(d_perstar or None),
                                               C.c_void_p(stream or None)))

    def mcmc_run_block(self, params, logpost, walker_ids, free, chol, seed, step0, n_steps, record=True):
        """Device-resident Metropolis block (b9_mcmc_run_block).  Same contract as
        mcmc.HostBlockRunner.run: returns (params, logpost, samples, lps, n_accept)."""
        return self.mcmc_collect(self.mcmc_submit(params, logpost, walker_ids, free, chol, seed, step0, n_steps, record,
                                                  cont=False, asynchronous=False))

    def mcmc_submit(self, params, logpost, walker_ids, free, chol, seed, step0, n_steps, record=True, cont=False,
                    asynchronous=True, row_origin=None):
        """Enqueue a block (B9_BLOCK_ASYNC) -- with cont=True from the state the previous block left on the device
        (B9_BLOCK_CONTINUE; params / logpost then only give the shapes).  Returns a handle for mcmc_collect; at
        most two handles may be outstanding and they are collected in submission order.  With `row_origin` the
        handle's "rows" receive the block's per-walker summary rows (b9_mcmc_block::rows)."""
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, abi.B9_NPARAM).copy()
        W, d = params.shape[0], len(free)
        logpost = np.ascontiguousarray(logpost, dtype=np.float64).copy()
        keep = dict(params=params, logpost=logpost, free=np.ascontiguousarray(free, dtype=np.int32),
                    ids=np.ascontiguousarray(walker_ids, dtype=np.int32), chol=np.ascontiguousarray(chol, dtype=np.float64),
                    samples=np.empty((n_steps, W, d)) if record else None, lps=np.empty((n_steps, W)) if record else None)
        blk = abi.b9_mcmc_block()
        blk.n_walkers, blk.n_free = W, d
        blk.free_idx = keep["free"].ctypes.data_as(_ip)
        blk.chol = keep["chol"].ctypes.data_as(_dp)
        blk.walker_ids = keep["ids"].ctypes.data_as(_ip)
        blk.seed, blk.step0, blk.n_steps = int(seed), int(step0), int(n_steps)
        blk.flags = (abi.BLOCK_CONTINUE if cont else 0) | (abi.BLOCK_ASYNC if asynchronous else 0)
        blk.params = params.ctypes.data_as(_dp)
        blk.logpost = logpost.ctypes.data_as(_dp)
        blk.samples = keep["samples"].ctypes.data_as(_dp) if record else None
        blk.lps = keep["lps"].ctypes.data_as(_dp) if record else None
        if row_origin is not None:      # the block's last launch also condenses every walker's chain into a summary row
            keep["origin"] = np.ascontiguousarray(row_origin, dtype=np.float64)
            keep["rows"] = np.empty((W, abi.row_doubles(d)))
            blk.row_origin = keep["origin"].ctypes.data_as(_dp)
            blk.rows = keep["rows"].ctypes.data_as(_dp)
        self._check(self.lib.b9_mcmc_run_block(self._ctx, C.byref(blk)))
        keep["blk"], keep["pending"] = blk, bool(asynchronous and n_steps > 0)
        return keep

    def mcmc_collect(self, h):
        if h["pending"]:
            self._check(self.lib.b9_mcmc_wait(self._ctx, C.byref(h["blk"])))
            h["pending"] = False
        return h["params"], h["logpost"], h["samples"], h["lps"], int(h["blk"].n_accept)

    def sample_mass(self, params: np.ndarray, seed: int = 1, row0: int = 0):
        """b9_sample_mass: per (row, star) one Gumbel-max draw of (primary mass, mass ratio, population) on the
        marginalisation grid + the membership probability.  Returns (mass, ratio, member, pop), each [rows, n_stars]."""
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, abi.B9_NPARAM)
        nr = params.shape[0]
        mass, ratio, member = (np.empty((nr, self.n_stars)) for _ in range(3))
        pop = np.empty((nr, self.n_stars), dtype=np.int32)
        self._check(self.lib.b9_sample_mass(self._ctx, params.ctypes.data_as(_dp), nr, int(seed), int(row0),
                                            mass.ctypes.data_as(_dp), ratio.ctypes.data_as(_dp), member.ctypes.data_as(_dp),
                                            pop.ctypes.data_as(C.POINTER(C.c_int32))))
        return mass, ratio, member, pop

    def derive_isochrone(self, param_row: np.ndarray, pop: int = 0, cap: int = 4096) -> Tuple[int, np.ndarray, np.ndarray, float]:
        row = np.ascontiguousarray(param_row, dtype=np.float64)
        mass = np.empty(cap)
        mags = np.empty(cap * self.n_filt)
        first, n, tip = C.c_int32(0), C.c_int32(0), C.c_double(0)
        self._check(self.lib.b9_derive_isochrone(self._ctx, row.ctypes.data_as(_dp), pop, cap,
                                                 mass.ctypes.data_as(_dp), mags.ctypes.data_as(_dp),
                                                 C.byref(first), C.byref(n), C.byref(tip)))
        k = n.value
        return first.value, mass[:k].copy(), mags[:k * self.n_filt].reshape(k, self.n_filt).copy(), tip.value

    # -- introspection --------------------------------------------------------------------
    def bytes_per_star_eval(self) -> int:
        return int(self.lib.b9_bytes_per_star_eval(self._ctx))

    def step_tiles_per_block(self, n_walkers: int) -> int:
        r = int(self.lib.b9_step_tiles_per_block(self._ctx, int(n_walkers)))
        if r < 0:
            self._check(r)
        return r

    def step_depth(self, n_walkers: int) -> int:
        r = int(self.lib.b9_step_depth(self._ctx, int(n_walkers)))
        if r < 0:
            self._check(r)
        return r

    def device_id(self) -> int:
        return int(self.lib.b9_device_id(self._ctx))

    def max_eep(self) -> int:
        return int(self.lib.b9_max_eep(self._ctx))

    def enable_timing(self, every: int = 1) -> None:
        """0/False: off; n: bracket every n-th launch of the dominant kernel with HIP events."""
        self._check(self.lib.b9_enable_timing(self._ctx, int(every)))

    def calibrate_timing(self) -> float:
        """ms an event bracket adds to a kernel's own duration (measured on an empty kernel)."""
        ms = C.c_double(0)
        self._check(self.lib.b9_calibrate_timing(self._ctx, C.byref(ms)))
        return ms.value

    def clock_stamp(self, which: int) -> None:
        """Enqueue the opening (0) / closing (1) shader-clock stamp on the context's stream."""
        self._check(self.lib.b9_clock_stamp(self._ctx, int(which)))

    def clock_mhz(self) -> Dict:
        """Shader clock between the two stamps: median over the compute units, extremes, length of the stretch (reference clock)."""
        m, lo, hi, sec = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
        self._check(self.lib.b9_clock_mhz(self._ctx, C.byref(m), C.byref(lo), C.byref(hi), C.byref(sec)))
        return {"mhz": m.value, "mhz_min_cu": lo.value, "mhz_max_cu": hi.value, "ref_seconds": sec.value}

    def kernel_time_ms(self, reset: bool = True) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_int32(0)
        self._check(self.lib.b9_kernel_time_ms(self._ctx, 1 if reset else 0, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def make_problem(pack_dict: Dict, cluster: Dict, n_pops: int = 1, mode: int = abi.MODE_GIVEN_MASS,
                 marg_iso_increm: int = 8, marg_n_q: int = 8):
    """Pin a synthetic pack + cluster into ABI structs: (pack, stars, priors, options)."""
    from . import synth
    pack = abi.make_pack(pack_dict)
    stars = abi.make_stars(cluster)
    priors = synth.default_priors(pack_dict, cluster["truth"], n_pops)
    options = abi.make_options(mode, n_pops, marg_iso_increm, marg_n_q)
    return pack, stars, priors, options
