"""ctypes binding of libbase9host.so (include/base9_host.h): the C++ walker-parallel sampler and its exchanges.

Plumbing only.  `HostSampler` drives the C++ `b9h::WalkerSampler` -- on the GPU through a `b9_ctx` (the product path:
device-resident blocks, summary rows condensed on the device, RCCL all-gather from HBM), or, for tests of the C++ host
logic on a machine without a GPU, through caller-supplied callbacks (block runner, evaluator, all-gather).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Sequence

import numpy as np

from . import abi

HOST_DIR = os.path.join(abi.REPO_ROOT, "base_amd", "host")
HOST_LIB_PATH = os.path.join(HOST_DIR, "libbase9host.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
vp = C.c_void_p

GATHER_FN = C.CFUNCTYPE(C.c_int, vp, _dp, C.c_size_t, _dp)
BLOCK_FN = C.CFUNCTYPE(C.c_int, vp, _dp, _dp, _ip, C.c_int, _ip, C.c_int, _dp, C.c_uint64, C.c_int64, C.c_int,
                       _dp, _dp, _dp, _dp, C.POINTER(C.c_int64))
LOGPOST_FN = C.CFUNCTYPE(C.c_int, vp, _dp, C.c_int, _dp)

#: every function include/base9_host.h declares (checked by tests/test_abi.py against the header)
HOST_SYMBOLS = [
    "b9h_last_error", "b9h_rank_from_env", "b9h_device_synchronize",
    "b9h_exchange_local", "b9h_exchange_rccl", "b9h_exchange_callback", "b9h_exchange_free", "b9h_exchange_barrier",
    "b9h_exchange_max", "b9h_exchange_world", "b9h_exchange_name", "b9h_exchange_comm_ranks", "b9h_exchange_devices",
    "b9h_group_check", "b9h_forced_ranks", "b9h_test_stall",
    "b9h_sampler_create", "b9h_sampler_create_callback", "b9h_sampler_free", "b9h_sampler_initialise", "b9h_sampler_run",
    "b9h_sampler_n_local", "b9h_sampler_state", "b9h_summary_rows",
    "b9h_load_pack", "b9h_free_pack", "b9h_read_phot", "b9h_free_phot", "b9h_settings_dump", "b9h_merge_parts",
]

_lib = None


def load() -> C.CDLL:
    """dlopen libbase9host.so (which pulls in libbase9hip.so, librccl and libamdhip64) and declare its prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError(f"{HOST_LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(HOST_LIB_PATH)
    lib.b9h_last_error.restype = C.c_char_p
    lib.b9h_rank_from_env.argtypes = [C.POINTER(C.c_int)] * 3
    lib.b9h_rank_from_env.restype = None
    lib.b9h_exchange_local.argtypes = [C.POINTER(vp)]
    lib.b9h_exchange_rccl.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.b9h_exchange_callback.argtypes = [GATHER_FN, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.b9h_exchange_free.argtypes = [vp]
    lib.b9h_exchange_free.restype = None
    lib.b9h_exchange_barrier.argtypes = [vp]
    lib.b9h_exchange_max.argtypes = [vp, C.c_double, _dp]
    lib.b9h_exchange_world.argtypes = [vp]
    lib.b9h_exchange_name.argtypes = [vp]
    lib.b9h_exchange_name.restype = C.c_char_p
    lib.b9h_exchange_comm_ranks.argtypes = [vp]
    lib.b9h_exchange_devices.argtypes = [vp, C.c_char_p, C.c_int]
    lib.b9h_test_stall.argtypes = [C.c_char_p, C.c_int]
    lib.b9h_group_check.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    lib.b9h_test_stall.restype = None
    lib.b9h_sampler_create.argtypes = [vp, C.c_int, C.c_int, _ip, _dp, C.c_int, C.c_uint64, C.c_int, vp, C.POINTER(vp)]
    lib.b9h_sampler_create_callback.argtypes = [BLOCK_FN, LOGPOST_FN, vp, C.c_int, _ip, _dp, C.c_int, C.c_uint64, C.c_int, vp, C.POINTER(vp)]
    lib.b9h_sampler_free.argtypes = [vp]
    lib.b9h_sampler_free.restype = None
    lib.b9h_sampler_initialise.argtypes = [vp, _dp]
    lib.b9h_sampler_run.argtypes = [vp, C.c_int64, C.c_int, _dp, _dp]
    lib.b9h_sampler_n_local.argtypes = [vp]
    lib.b9h_sampler_state.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _dp, _dp, _dp, _dp]
    lib.b9h_summary_rows.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp]
    lib.b9h_load_pack.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(vp), C.POINTER(abi.b9_pack)]
    lib.b9h_free_pack.argtypes = [vp]
    lib.b9h_read_phot.argtypes = [C.c_char_p, C.c_double, C.c_double, C.c_int, C.POINTER(vp), C.POINTER(abi.b9_stars), C.c_char_p, C.c_int]
    lib.b9h_free_phot.argtypes = [vp]
    lib.b9h_settings_dump.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
    lib.b9h_merge_parts.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_long]
    _lib = lib
    return lib


class HostError(RuntimeError):
    pass


def _check(rc: int) -> None:
    if rc != 0:
        raise HostError(load().b9h_last_error().decode())


def rank_from_env():
    r, w, l = C.c_int(0), C.c_int(1), C.c_int(0)
    load().b9h_rank_from_env(C.byref(r), C.byref(w), C.byref(l))
    return r.value, w.value, l.value


class Exchange:
    """How the walkers' block summary rows travel between ranks."""

    def __init__(self, handle: vp, keep=None):
        self._h, self._keep = handle, keep

    @classmethod
    def local(cls) -> "Exchange":
        h = vp()
        _check(load().b9h_exchange_local(C.byref(h)))
        return cls(h)

    @classmethod
    def rccl(cls, rank: int, world: int, device: int, directory: Optional[str] = None) -> "Exchange":
        """One process per GPU; the RCCL unique id travels through a file (b9dist.hpp)."""
        h = vp()
        _check(load().b9h_exchange_rccl(rank, world, directory.encode() if directory else None, device, C.byref(h)))
        return cls(h)

    @classmethod
    def callback(cls, all_gather: Callable[[np.ndarray], np.ndarray], rank: int, world: int) -> "Exchange":
        """Test seam: `all_gather(rows[count]) -> rows[world * count]` (e.g. gloo through torch.distributed)."""
        def _gather(_user, mine, count, out):
            try:
                got = np.asarray(all_gather(np.ctypeslib.as_array(mine, shape=(count,)).copy()), dtype=np.float64).ravel()
                np.ctypeslib.as_array(out, shape=(count * world,))[:] = got
                return 0
            except Exception:      # noqa: BLE001 -- reported through the C return code
                import traceback
                traceback.print_exc()
                return 1
        fn = GATHER_FN(_gather)
        h = vp()
        _check(load().b9h_exchange_callback(fn, None, rank, world, C.byref(h)))
        return cls(h, keep=fn)

    def barrier(self) -> None:
        _check(load().b9h_exchange_barrier(self._h))

    def max(self, value: float) -> float:
        out = C.c_double(0)
        _check(load().b9h_exchange_max(self._h, float(value), C.byref(out)))
        return out.value

    @property
    def world(self) -> int:
        return int(load().b9h_exchange_world(self._h))

    @property
    def name(self) -> str:
        return load().b9h_exchange_name(self._h).decode()

    @property
    def comm_ranks(self) -> int:
        """ncclCommCount of the exchange's communicator (0: it has none)."""
        return int(load().b9h_exchange_comm_ranks(self._h))

    @property
    def devices(self):
        """PCI bus ids of the ranks' GPUs in rank order, gathered through the communicator ([] without one)."""
        buf = C.create_string_buffer(4096)
        if load().b9h_exchange_devices(self._h, buf, len(buf)) != 0:
            raise HostError("b9h_exchange_devices: buffer too small")
        txt = buf.value.decode()
        return txt.split(",") if txt else []

    def close(self) -> None:
        if self._h:
            load().b9h_exchange_free(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:     # noqa: BLE001
            pass


class HostSampler:
    """b9h::WalkerSampler.  `engine` = a base_amd.engine.Engine (GPU), or runner callbacks for the CPU test seam."""

    def __init__(self, n_walkers: int, free: Sequence[int], step: Sequence[float], exchange: Exchange, seed: int = 1234,
                 block: int = 50, engine=None, run_block: Optional[Callable] = None, evaluate: Optional[Callable] = None):
        lib = load()
        self.exchange = exchange
        self.free = np.ascontiguousarray(free, dtype=np.int32)
        self.d = len(self.free)
        self.n_walkers = int(n_walkers)
        stepv = np.ascontiguousarray(step, dtype=np.float64)
        self._h = vp()
        self._keep = []
        if engine is not None:
            self._keep.append(engine)
            _check(lib.b9h_sampler_create(engine._ctx, int(engine.options.mode), self.n_walkers, self.free.ctypes.data_as(_ip),
                                          stepv.ctypes.data_as(_dp), self.d, int(seed), int(block), exchange._h, C.byref(self._h)))
        else:
            d = self.d

            def _run(_u, p_in, lp_in, ids, n_local, free_idx, dd, chol, seed_, step0, n_steps, p_out, lp_out, samples, lps, n_acc):
                try:
                    params = np.ctypeslib.as_array(p_in, shape=(n_local, abi.B9_NPARAM)).copy()
                    logpost = np.ctypeslib.as_array(lp_in, shape=(n_local,)).copy()
                    wid = np.ctypeslib.as_array(ids, shape=(n_local,)).copy()
                    fr = np.ctypeslib.as_array(free_idx, shape=(dd,)).astype(np.int64)
                    ch = np.ctypeslib.as_array(chol, shape=(dd, dd)).copy()
                    po, lo, sa, ls, na = run_block(params, logpost, wid, fr, ch, int(seed_), int(step0), int(n_steps))
                    np.ctypeslib.as_array(p_out, shape=(n_local, abi.B9_NPARAM))[:] = po
                    np.ctypeslib.as_array(lp_out, shape=(n_local,))[:] = lo
                    np.ctypeslib.as_array(samples, shape=(n_steps, n_local, dd))[:] = sa
                    np.ctypeslib.as_array(lps, shape=(n_steps, n_local))[:] = ls
                    n_acc[0] = int(na)
                    return 0
                except Exception:      # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1

            def _eval(_u, params, n, out):
                try:
                    np.ctypeslib.as_array(out, shape=(n,))[:] = evaluate(np.ctypeslib.as_array(params, shape=(n, abi.B9_NPARAM)).copy())
                    return 0
                except Exception:      # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1

            self._keep += [BLOCK_FN(_run), LOGPOST_FN(_eval)]
            _check(lib.b9h_sampler_create_callback(self._keep[0], self._keep[1], None, self.n_walkers, self.free.ctypes.data_as(_ip),
                                                   stepv.ctypes.data_as(_dp), d, int(seed), int(block), exchange._h, C.byref(self._h)))
        self.n_local = int(lib.b9h_sampler_n_local(self._h))

    def initialise(self, start: np.ndarray) -> None:
        start = np.ascontiguousarray(start, dtype=np.float64).reshape(self.n_walkers, abi.B9_NPARAM)
        _check(load().b9h_sampler_initialise(self._h, start.ctypes.data_as(_dp)))

    def run(self, n_steps: int, adapt: bool = True, record: bool = False):
        """Returns (samples[n_steps, n_local, d], lps[n_steps, n_local]) when `record`, else None."""
        if record:
            samples = np.empty((n_steps, self.n_local, self.d))
            lps = np.empty((n_steps, self.n_local))
            _check(load().b9h_sampler_run(self._h, int(n_steps), int(bool(adapt)), samples.ctypes.data_as(_dp), lps.ctypes.data_as(_dp)))
            return samples, lps
        _check(load().b9h_sampler_run(self._h, int(n_steps), int(bool(adapt)), None, None))
        return None

    def state(self) -> dict:
        steps, acc, scale = C.c_int64(0), C.c_int64(0), C.c_double(0)
        chol = np.empty((self.d, self.d))
        allp = np.empty((self.n_walkers, abi.B9_NPARAM))
        alll = np.empty(self.n_walkers)
        _check(load().b9h_sampler_state(self._h, C.byref(steps), C.byref(acc), C.byref(scale), chol.ctypes.data_as(_dp),
                                        allp.ctypes.data_as(_dp), alll.ctypes.data_as(_dp)))
        return dict(steps=steps.value, accepted_local=acc.value, scale=scale.value, chol=chol, all_params=allp, all_logpost=alll)

    def close(self) -> None:
        if self._h:
            load().b9h_sampler_free(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:     # noqa: BLE001
            pass


def summary_rows(samples: np.ndarray, params_end: np.ndarray, logpost_end: np.ndarray, origin: np.ndarray) -> np.ndarray:
    """b9h::summary_rows: the host statement of the device's block summary rows."""
    samples = np.ascontiguousarray(samples, dtype=np.float64)
    n, wl, d = samples.shape
    pe = np.ascontiguousarray(params_end, dtype=np.float64)
    le = np.ascontiguousarray(logpost_end, dtype=np.float64)
    og = np.ascontiguousarray(origin, dtype=np.float64)
    rows = np.empty((wl, abi.row_doubles(d)))
    _check(load().b9h_summary_rows(samples.ctypes.data_as(_dp), pe.ctypes.data_as(_dp), le.ctypes.data_as(_dp), n, wl, d,
                                   og.ctypes.data_as(_dp), rows.ctypes.data_as(_dp)))
    return rows
