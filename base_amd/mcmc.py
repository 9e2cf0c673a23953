"""Walker-parallel adaptive Metropolis over the HIP log-posterior (BASELINE.json north_star:
"partition independent walkers/chains across the GPUs of one node with an RCCL all-gather of
log-posteriors for the adaptive proposal step").

Design (there is no reference counterpart -- the reference runs one chain on CPU
threads [RECALL], "walkers" are a construct of the build contract):

  * W independent Metropolis chains ("walkers"); walker w lives on rank  w // (W / world).
  * Every rank holds the full ensemble state (W x B9_NPARAM doubles -- a few KB) and draws ALL
    walkers' proposals from counter-based per-walker Philox streams, so proposals are identical
    on every rank and independent of the number of GPUs.
  * Each rank evaluates only its own walkers' log-posteriors on its GPU (b9_logpost_device,
    device-resident), then ONE collective per step -- an all-gather of the W log-posteriors
    (RCCL over xGMI on GPUs; gloo in the CPU tests) -- gives every rank all of them.
  * Accept/reject and the adaptive step (pooled proposal covariance over all walkers, as in
    the reference's staged burn-in adaptation [RECALL]) are then replicated on every rank.

The only data-path traffic between GPUs is that 8*W-byte all-gather; it is latency-bound.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from . import abi

DEFAULT_FREE = (abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS)
DEFAULT_STEP = {abi.P_LOGAGE: 0.005, abi.P_Y: 0.003, abi.P_FEH: 0.01, abi.P_MOD: 0.005, abi.P_ABS: 0.003,
                abi.P_CARBONICITY: 0.01, abi.P_IFMR_INTERCEPT: 0.005, abi.P_IFMR_SLOPE: 0.005,
                abi.P_IFMR_QUAD: 0.002, abi.P_Y2: 0.003, abi.P_LAMBDA: 0.01}


class Ensemble:
    """Replicated ensemble state + proposal machinery (pure numpy; identical on every rank)."""

    def __init__(self, start: np.ndarray, free: Sequence[int] = DEFAULT_FREE, seed: int = 1234,
                 adapt_start: int = 200, adapt_every: int = 100, step_sizes: Optional[dict] = None):
        self.params = np.ascontiguousarray(start, dtype=np.float64).reshape(-1, abi.B9_NPARAM).copy()
        self.n_walkers = self.params.shape[0]
        self.free = np.array(list(free), dtype=np.int64)
        self.d = len(self.free)
        self.logpost = np.full(self.n_walkers, -np.inf)
        self.step = 0
        ss = dict(DEFAULT_STEP)
        if step_sizes:
            ss.update(step_sizes)
        self.chol = np.diag([ss[int(k)] for k in self.free])
        self.adapt_start, self.adapt_every = adapt_start, adapt_every
        # one counter-based stream per walker: the same draws whatever the rank layout
        self.rng = [np.random.Generator(np.random.Philox(key=[seed, w])) for w in range(self.n_walkers)]
        # running pooled moments of the accepted states (for the adaptive covariance)
        self.n_mom = 0
        self.mean = np.zeros(self.d)
        self.m2 = np.zeros((self.d, self.d))
        self.accepted = 0
        self._prop = self.params.copy()
        self._u = np.zeros(self.n_walkers)

    def propose(self) -> np.ndarray:
        z = np.empty((self.n_walkers, self.d))
        for w, g in enumerate(self.rng):
            z[w] = g.standard_normal(self.d)
            self._u[w] = g.random()
        self._prop[:] = self.params
        self._prop[:, self.free] += z @ self.chol.T
        return self._prop

    def accept(self, logpost_prop: np.ndarray) -> np.ndarray:
        with np.errstate(invalid="ignore"):
            ok = np.log(self._u) < (logpost_prop - self.logpost)
        ok &= np.isfinite(logpost_prop)
        self.params[ok] = self._prop[ok]
        self.logpost[ok] = logpost_prop[ok]
        self.accepted += int(ok.sum())
        self.step += 1
        # pooled running covariance (Welford, one update per walker per step)
        x = self.params[:, self.free]
        for w in range(self.n_walkers):
            self.n_mom += 1
            dlt = x[w] - self.mean
            self.mean += dlt / self.n_mom
            self.m2 += np.outer(dlt, x[w] - self.mean)
        if self.step >= self.adapt_start and self.step % self.adapt_every == 0 and self.n_mom > 10 * self.d:
            cov = self.m2 / (self.n_mom - 1)
            cov = cov * (2.38 ** 2 / self.d) + 1e-12 * np.eye(self.d)
            try:
                self.chol = np.linalg.cholesky(cov)
            except np.linalg.LinAlgError:
                pass
        return ok


class WalkerSampler:
    """Drives an Ensemble with a log-posterior evaluator sharded over torch.distributed ranks.

    `evaluate_local(params_local) -> logpost_local` is either the GPU path (DeviceEvaluator) or,
    in the gloo CPU tests, any callable on numpy arrays.
    """

    def __init__(self, ensemble: Ensemble, evaluate_local: Callable, rank: int = 0, world: int = 1,
                 gather: Optional[Callable] = None):
        if ensemble.n_walkers % world:
            raise ValueError("the number of walkers must be a multiple of the number of ranks")
        self.ens, self.rank, self.world = ensemble, rank, world
        self.per = ensemble.n_walkers // world
        self.lo, self.hi = rank * self.per, (rank + 1) * self.per
        self.evaluate_local = evaluate_local
        self.gather = gather

    def initialise(self) -> None:
        self.ens.logpost[:] = self._eval(self.ens.params)

    def _eval(self, params_all: np.ndarray) -> np.ndarray:
        local = self.evaluate_local(params_all[self.lo:self.hi])
        if self.world == 1:
            return np.asarray(local, dtype=np.float64).copy()
        return self.gather(local)

    def step(self) -> np.ndarray:
        prop = self.ens.propose()
        lp = self._eval(prop)
        return self.ens.accept(lp)

    def run(self, n_steps: int, record: Optional[List] = None) -> None:
        for _ in range(n_steps):
            self.step()
            if record is not None:
                record.append((self.ens.params.copy(), self.ens.logpost.copy()))


class DeviceEvaluator:
    """GPU evaluation of the local walkers + RCCL all-gather, all on torch's current stream.

    torch is plumbing here: it owns the device buffers and the process group; the numbers come
    from b9_logpost_device (hand-written HIP behind the C ABI).
    """

    def __init__(self, engine, n_local: int, world: int = 1, device: Optional[int] = None):
        import torch
        self.torch = torch
        self.engine = engine
        self.world = world
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.h_params = torch.empty((n_local, abi.B9_NPARAM), dtype=torch.float64).pin_memory()
        self.d_params = torch.empty((n_local, abi.B9_NPARAM), dtype=torch.float64, device=dev)
        self.d_local = torch.empty(n_local, dtype=torch.float64, device=dev)
        self.d_all = torch.empty(n_local * world, dtype=torch.float64, device=dev)
        self.h_all = torch.empty(n_local * world, dtype=torch.float64).pin_memory()
        self.n_local = n_local

    def evaluate_and_gather(self, params_local: np.ndarray) -> np.ndarray:
        torch = self.torch
        self.h_params.numpy()[:] = params_local
        self.d_params.copy_(self.h_params, non_blocking=True)
        stream = torch.cuda.current_stream().cuda_stream
        self.engine.logpost_device(self.d_params.data_ptr(), self.n_local, self.d_local.data_ptr(), 0, stream)
        if self.world > 1:
            torch.distributed.all_gather_into_tensor(self.d_all, self.d_local)
            src = self.d_all
        else:
            src = self.d_local
        self.h_all.copy_(src, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.h_all.numpy().copy()


def make_device_sampler(engine, start: np.ndarray, rank: int, world: int, **ens_kw) -> WalkerSampler:
    ens = Ensemble(start, **ens_kw)
    per = ens.n_walkers // world
    ev = DeviceEvaluator(engine, per, world)
    s = WalkerSampler(ens, ev.evaluate_and_gather, rank, world, gather=lambda x: x)
    # evaluate_and_gather already returns the gathered vector; bypass the second gather
    s._eval = lambda params_all: ev.evaluate_and_gather(params_all[s.lo:s.hi])   # noqa: E731
    s.evaluator = ev
    return s
