"""Python twin of the walker-parallel driver (the product driver is the C++ one, base_amd/host/b9sampler.cpp, bound by
base_amd/hostlib.py; this module holds the numpy statement of the device's Metropolis step -- HostBlockRunner, draws --
that the tests compare the GPU against, and a torch.distributed flavour of the same sampler used by the gloo tests).

Walker-parallel adaptive Metropolis over the HIP log-posterior (BASELINE.json north_star:
"partition independent walkers/chains across the GPUs of one node with an RCCL all-gather of
log-posteriors over xGMI for the adaptive proposal step").

Design (there is no reference counterpart -- the reference runs one chain on CPU
threads [RECALL]; "walkers" are a construct of the build contract):

  * W independent Metropolis chains ("walkers"); walker w lives on rank  w // (W / world).
  * Between adaptation points a walker needs nothing from any other walker: every rank advances
    its own walkers for `block` steps with no communication at all (on a GPU the whole block is
    device-resident: see DeviceBlockRunner).
  * At the adaptive-proposal step -- once per block -- every rank contributes one row per local
    walker, [log-posterior, position, block moments], to ONE all-gather (RCCL over xGMI on
    GPUs, gloo in the CPU tests).  Every rank then pools the rows in walker order and derives the
    same proposal covariance (adaptive Metropolis, 2.38^2/d scaling), as the reference's staged
    burn-in adaptation does for its single chain [RECALL].
  * Random numbers are counter-based (Philox4x32-10, key = seed, counter = (step, walker, draw)):
    a walker's chain is the same whatever the number of ranks, and the device runner reproduces
    the host reference draw for draw.

The all-gather moves (15 + 2d + d^2) doubles per walker per block -- a few hundred bytes; it is
latency-bound, and amortised over `block` steps.
"""
from __future__ import annotations

import concurrent.futures
import os

from typing import Callable, List, Optional, Sequence

import numpy as np

from . import abi

DEFAULT_FREE = (abi.P_LOGAGE, abi.P_FEH, abi.P_MOD, abi.P_ABS)
DEFAULT_STEP = {abi.P_LOGAGE: 5e-4, abi.P_Y: 3e-4, abi.P_FEH: 1e-3, abi.P_MOD: 5e-4, abi.P_ABS: 3e-4,
                abi.P_CARBONICITY: 1e-3, abi.P_IFMR_INTERCEPT: 5e-4, abi.P_IFMR_SLOPE: 5e-4,
                abi.P_IFMR_QUAD: 2e-4, abi.P_Y2: 3e-4, abi.P_LAMBDA: 1e-3}

# ------------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al. 2011), vectorised; the device runner implements the same function
# ------------------------------------------------------------------------------------------
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1):
    """10 rounds; all arguments uint32 arrays (broadcastable).  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _W0); k1 = np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def _u01(hi, lo):
    """(0,1) double from two uint32: 53 random bits, never 0 or 1."""
    x = (hi.astype(np.uint64) >> np.uint64(5)) * np.uint64(1 << 26) + (lo.astype(np.uint64) >> np.uint64(6))
    return (x.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def draws(seed: int, step: int, walkers: np.ndarray, d: int):
    """Standard normals z[len(walkers), d] and uniforms u[len(walkers)] for one step."""
    walkers = np.asarray(walkers, dtype=np.uint32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    s_lo, s_hi = np.uint32(step & 0xFFFFFFFF), np.uint32((step >> 32) & 0xFFFFFFFF)
    z = np.empty((len(walkers), d))
    n_pairs = (d + 1) // 2
    for j in range(n_pairs):                         # Box-Muller, one Philox call per pair
        r = philox4x32(s_lo, s_hi, walkers, np.uint32(j), k0, k1)
        u1, u2 = _u01(r[0], r[1]), _u01(r[2], r[3])
        rad = np.sqrt(-2.0 * np.log(u1))
        z[:, 2 * j] = rad * np.cos(2.0 * np.pi * u2)
        if 2 * j + 1 < d:
            z[:, 2 * j + 1] = rad * np.sin(2.0 * np.pi * u2)
    r = philox4x32(s_lo, s_hi, walkers, np.uint32(n_pairs), k0, k1)
    return z, _u01(r[0], r[1])


# ------------------------------------------------------------------------------------------
# block runners: advance the LOCAL walkers `n_steps` steps with a fixed proposal factor
# ------------------------------------------------------------------------------------------
class HostBlockRunner:
    """Reference runner: proposals and accept/reject in numpy, log-posteriors from `evaluate`
    (the CPU oracle in tests, or the GPU engine's synchronous logpost)."""

    def __init__(self, evaluate: Callable[[np.ndarray], np.ndarray]):
        self.evaluate = evaluate

    def run(self, params, logpost, walker_ids, free, chol, seed, step0, n_steps):
        """Returns (params, logpost, samples[n_steps, Wl, d], lps[n_steps, Wl], n_accept)."""
        params, logpost = params.copy(), logpost.copy()
        d = len(free)
        samples = np.empty((n_steps, len(walker_ids), d))
        lps = np.empty((n_steps, len(walker_ids)))
        n_acc = 0
        for s in range(n_steps):
            z, u = draws(seed, step0 + s, walker_ids, d)
            prop = params.copy()
            delta = np.zeros_like(z)                 # delta_i = sum_j chol[i, j] z_j, j ascending,
            for j in range(d):                       # plain multiply-add: the device does the same
                delta = delta + chol[None, :, j] * z[:, j:j + 1]
            prop[:, free] += delta
            lp = np.asarray(self.evaluate(prop), dtype=np.float64)
            with np.errstate(invalid="ignore"):
                ok = (np.log(u) < lp - logpost) & np.isfinite(lp)
            params[ok] = prop[ok]
            logpost[ok] = lp[ok]
            n_acc += int(ok.sum())
            samples[s] = params[:, free]
            lps[s] = logpost
        return params, logpost, samples, lps, n_acc


class DeviceBlockRunner:
    """The whole block on the GPU (b9_mcmc_run_block): one launch per step in given-mass mode (the
    fused step: previous step's accept/reject + this step's star likelihood + both candidate
    isochrone sets of the next step), two in marginalised mode; no host round trip until the block ends."""

    def __init__(self, engine, record: bool = True):
        self.engine, self.record = engine, record

    def run(self, params, logpost, walker_ids, free, chol, seed, step0, n_steps):
        return self.engine.mcmc_run_block(params, logpost, walker_ids, free, chol, seed, step0, n_steps, self.record)

    # Pipelining (given-mass mode): a block can be enqueued -- continuing from the device-resident state of the
    # block before it -- while that block still runs, so the GPU's queue never drains between blocks.
    def can_pipeline(self) -> bool:
        return self.engine.options.mode == abi.MODE_GIVEN_MASS and not os.environ.get("B9_TWO_LAUNCH_STEPS") \
            and not os.environ.get("B9_NO_BLOCK_PIPELINE")

    def submit(self, params, logpost, walker_ids, free, chol, seed, step0, n_steps, cont):
        return self.engine.mcmc_submit(params, logpost, walker_ids, free, chol, seed, step0, n_steps, self.record, cont=cont)

    def collect(self, handle):
        return self.engine.mcmc_collect(handle)


FORGET = 0.9          # per-block forgetting factor of the pooled adaptation moments
SCALE_MIN, SCALE_MAX = 1e-8, 1e8     # the global step scale stays finite whatever the acceptance does (flat or improper posteriors)


class WalkerSampler:
    """W walkers sharded over `world` ranks; one all-gather per adaptation block."""

    def __init__(self, start: np.ndarray, runner, rank: int = 0, world: int = 1,
                 all_gather: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                 free: Sequence[int] = DEFAULT_FREE, seed: int = 1234, block: int = 50,
                 step_sizes: Optional[dict] = None, adapt: bool = True):
        start = np.ascontiguousarray(start, dtype=np.float64).reshape(-1, abi.B9_NPARAM)
        self.n_walkers = start.shape[0]
        if self.n_walkers % world:
            raise ValueError("the number of walkers must be a multiple of the number of ranks")
        self.rank, self.world, self.runner = rank, world, runner
        self.per = self.n_walkers // world
        self.ids = np.arange(rank * self.per, (rank + 1) * self.per)
        self.free = np.array(list(free), dtype=np.int64)
        self.d = len(self.free)
        self.seed, self.block, self.adapt = seed, block, adapt
        ss = dict(DEFAULT_STEP)
        if step_sizes:
            ss.update(step_sizes)
        self.chol = np.diag([ss[int(k)] for k in self.free]).astype(np.float64)
        self.scale = 1.0                              # global step scale, steered by the pooled acceptance
        self.params = start[self.ids].copy()          # local walkers only
        self.logpost = np.full(self.per, -np.inf)
        self.all_params = start.copy()                # refreshed at every gather
        self.all_logpost = np.full(self.n_walkers, -np.inf)
        self.step = 0
        self.accepted = 0
        self._gather = all_gather
        self._pending = None                          # (gather in flight, its block length)
        # pooled running sums over all walkers and all blocks, about the common origin x0
        self.x0 = start[:, self.free].mean(axis=0)
        self.n_mom, self.s1, self.s2 = 0.0, np.zeros(self.d), np.zeros((self.d, self.d))
        self._shaped = False                          # True once a learnt covariance has replaced the diagonal steps

    # -- collectives ------------------------------------------------------------------------
    def gather_rows(self, rows: np.ndarray) -> np.ndarray:
        """rows[per, k] on every rank -> [n_walkers, k] in walker order (blocking)."""
        return self._finish_gather(self._start_gather(rows))

    def _start_gather(self, rows: np.ndarray):
        if self._gather is None:
            if self.world != 1:
                raise ValueError("more than one rank needs an all_gather")
            return rows.copy()
        start = getattr(self._gather, "start", None)
        return start(rows) if start else self._gather(rows)

    def _finish_gather(self, pending) -> np.ndarray:
        return pending.wait() if hasattr(pending, "wait") else pending

    # -- driver -----------------------------------------------------------------------------
    def initialise(self, evaluate: Callable[[np.ndarray], np.ndarray]) -> None:
        self.logpost = np.asarray(evaluate(self.params), dtype=np.float64).copy()
        rows = self.gather_rows(np.concatenate([self.logpost[:, None], self.params], axis=1))
        self.all_logpost, self.all_params = rows[:, 0].copy(), rows[:, 1:].copy()

    def run_block(self, n_steps: Optional[int] = None):
        """One block, unpipelined: execute, start the block's gather, consume the previous block's."""
        n = self.block if n_steps is None else n_steps
        self.params, self.logpost, samples, lps, n_acc = self.runner.run(
            self.params, self.logpost, self.ids, self.free, self.scale * self.chol, self.seed, self.step, n)
        self.step += n
        self.accepted += n_acc
        row = self._make_row(samples, self.params, self.logpost, n)
        # THE collective of the block is started now and consumed after the NEXT block has run, so its
        # latency hides behind that block's GPU work: the proposal of block b+1 is adapted from the rows
        # of blocks <= b-1.  The lag is the same for every rank count (also for one rank), so chains
        # stay bit-identical however the walkers are sharded.
        previous, self._pending = self._pending, (self._start_gather(row), n)
        if previous is not None:
            self._consume(*previous)
        return samples, lps

    def _make_row(self, samples: np.ndarray, params_end: np.ndarray, logpost_end: np.ndarray, n: int) -> np.ndarray:
        """One row per local walker: [lp, full position, #moves, n, sum x, sum x x^T] over the block, x measured
        from the common origin x0 (the ensemble's starting mean) so that the pooled second moments do not
        cancel catastrophically."""
        x = samples - self.x0                                     # [n, per, d]
        moved = (np.abs(np.diff(samples, axis=0)).sum(axis=2) > 0).sum(axis=0)
        return np.concatenate([logpost_end[:, None], params_end, moved[:, None].astype(np.float64),
                               np.full((self.per, 1), float(n)), x.sum(axis=0), self._second_moments(x)], axis=1)

    @staticmethod
    def _second_moments(x: np.ndarray) -> np.ndarray:
        """sum over steps of x x^T per walker, flattened: [n, per, d] -> [per, d*d] (batched matmul: an order of
        magnitude cheaper than the equivalent einsum for these shapes)."""
        xw = np.ascontiguousarray(x.transpose(1, 0, 2))                  # [per, n, d]
        return (xw.transpose(0, 2, 1) @ xw).reshape(x.shape[1], -1)

    def _consume(self, pending, n: int) -> None:
        rows = self._finish_gather(pending)
        self.all_logpost = rows[:, 0].copy()
        self.all_params = rows[:, 1:1 + abi.B9_NPARAM].copy()
        if self.adapt:
            # pooled fraction of steps (after the block's first) on which a walker moved
            rate = rows[:, 1 + abi.B9_NPARAM].sum() / max(1.0, self.n_walkers * (n - 1.0))
            if n > 4:
                self.scale = min(max(self.scale * step_scale_factor(rate), SCALE_MIN), SCALE_MAX)
            self._adapt(rows[:, 2 + abi.B9_NPARAM:])

    def flush(self) -> None:
        """Consume the outstanding gather (end of a run): brings all_params / all_logpost up to date."""
        if self._pending is not None:
            previous, self._pending = self._pending, None
            self._consume(*previous)

    def _adapt(self, mom: np.ndarray) -> None:
        """Pool the block's per-walker sums (every rank sees the same gathered rows in the same
        order, so every rank derives the same proposal factor).  O(1) numpy calls per block,
        whatever the number of walkers."""
        d = self.d
        # exponentially forgotten sums (window ~ 1/(1-FORGET) blocks): the start-up transient and the part
        # of a degeneracy ridge the ensemble has already left stop shaping the proposal
        self.n_mom = FORGET * self.n_mom + float(mom[:, 0].sum())
        self.s1 = FORGET * self.s1 + mom[:, 1:1 + d].sum(axis=0)
        self.s2 = FORGET * self.s2 + mom[:, 1 + d:].sum(axis=0).reshape(d, d)
        if self.n_mom > 20 * d:
            mean = self.s1 / self.n_mom
            cov = (self.s2 - self.n_mom * np.outer(mean, mean)) / (self.n_mom - 1) * (2.38 ** 2 / d)
            # A chain that has hardly moved yet (bad starting scale) has a collapsed sample covariance:
            # adopting it would freeze the sampler.  Only take it once no direction is more than 100x
            # narrower than the current (scaled) proposal already is.  Same rule as the C++ driver.
            cur = self.scale ** 2 * np.einsum("ij,ij->i", self.chol, self.chol)
            if np.all(np.diag(cov) > 1e-4 * cur):
                try:
                    new = np.linalg.cholesky(cov * (1.0 + 1e-9 * np.eye(d)))
                except np.linalg.LinAlgError:
                    return
                if not self._shaped:
                    # first switch from the diagonal start-up steps to a learnt shape: keep the volume of
                    # the scaled proposal (the acceptance-tuned size carries over, only the shape changes)
                    self.scale *= float(np.exp(np.mean(np.log(np.diag(self.chol)) - np.log(np.diag(new)))))
                    self._shaped = True
                self.chol = new                                      # `scale` keeps multiplying it

    def run(self, n_steps: int, record: Optional[List] = None, adapt: Optional[bool] = None) -> None:
        """adapt = False freezes the proposal for this call (the main run after a burn-in, as the C++ driver and the
        reference's staged burn-in do [RECALL]); None keeps the constructor's setting.  A run() drains the pipeline when
        it returns, so run(a); run(b) adapts at a different point than run(a + b) and gives (equally valid) different
        chains.

        n_steps in blocks, PIPELINED: while block b executes (the runner's C call releases the GIL; a worker
        thread makes it), this thread turns block b-1's samples into rows, exchanges them (the all-gather then
        runs beside block b's kernels) and adapts the proposal.  Data dependencies are those of run_block --
        the proposal of block b+1 is adapted from the rows of blocks <= b-1 -- so the chains are the same bits."""
        self.flush()
        if adapt is not None:
            keep, self.adapt = self.adapt, adapt
            try:
                return self.run(n_steps, record)
            finally:
                self.adapt = keep
        if getattr(self.runner, "can_pipeline", lambda: False)() and n_steps > 0:
            return self._run_device_pipeline(n_steps, record)
        done, finished = 0, None            # finished: (samples, params_end, logpost_end, n) of the block that ran last
        with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:
            while done < n_steps:
                n = min(self.block, n_steps - done)
                fut = pool.submit(self.runner.run, self.params, self.logpost, self.ids, self.free,
                                  self.scale * self.chol, self.seed, self.step, n)
                if finished is not None:                             # overlapped with the block now running
                    s_prev, p_prev, l_prev, n_prev = finished
                    self._consume(self._start_gather(self._make_row(s_prev, p_prev, l_prev, n_prev)), n_prev)
                self.params, self.logpost, samples, lps, n_acc = fut.result()
                self.step += n
                self.accepted += n_acc
                if record is not None:
                    record.append((samples, lps))
                finished = (samples, self.params, self.logpost, n)
                done += n
        if finished is not None:
            s_prev, p_prev, l_prev, n_prev = finished
            self._consume(self._start_gather(self._make_row(s_prev, p_prev, l_prev, n_prev)), n_prev)


    def _run_device_pipeline(self, n_steps: int, record: Optional[List]) -> None:
        """Two blocks outstanding on the device: block b+1 is enqueued (continuing from block b's device-resident
        state) as soon as block b-1 has been collected and its rows exchanged and consumed -- i.e. with exactly the
        proposal the unpipelined loop would use -- while block b is still running."""
        sizes, left = [], n_steps
        while left > 0:
            sizes.append(min(self.block, left)); left -= sizes[-1]
        step = self.step
        handles = []

        def enqueue(b):
            nonlocal step
            handles.append(self.runner.submit(self.params, self.logpost, self.ids, self.free, self.scale * self.chol,
                                              self.seed, step, sizes[b], cont=b > 0))
            step += sizes[b]

        def finish(b):           # collect block b, account for it, exchange its rows, adapt
            self.params, self.logpost, samples, lps, n_acc = self.runner.collect(handles[b])
            handles[b] = None
            self.step += sizes[b]
            self.accepted += n_acc
            if record is not None:
                record.append((samples, lps))
            self._consume(self._start_gather(self._make_row(samples, self.params, self.logpost, sizes[b])), sizes[b])

        enqueue(0)
        for b in range(len(sizes)):
            if b >= 1:
                finish(b - 1)                                    # while block b runs
            if b + 1 < len(sizes):
                enqueue(b + 1)                                   # proposal adapted from blocks <= b-1, as unpipelined
        finish(len(sizes) - 1)


# ------------------------------------------------------------------------------------------
# torch.distributed plumbing
# ------------------------------------------------------------------------------------------
def step_scale_factor(rate: float) -> float:
    """Multiplicative update of the global step scale from a block's acceptance rate (target
    0.2-0.35; far from it the correction is strong).  Same rule as the C++ driver (b9host.cpp)."""
    if rate < 0.02: return 0.2
    if rate < 0.10: return 0.5
    if rate < 0.20: return 0.8
    if rate > 0.90: return 4.0
    if rate > 0.70: return 2.0
    if rate > 0.50: return 1.5
    if rate > 0.35: return 1.2
    return 1.0


class _PendingGather:
    def __init__(self, work, out, host=None):
        self.work, self.out, self.host = work, out, host

    def wait(self) -> np.ndarray:
        self.work.wait()
        if self.host is None:
            return self.out.cpu().numpy()
        self.host.copy_(self.out, non_blocking=True)      # pinned: no staging copy, one stream sync
        import torch
        torch.cuda.current_stream().synchronize()
        return self.host.numpy().copy()


def torch_all_gather(device: Optional[str] = None) -> Callable[[np.ndarray], np.ndarray]:
    """all-gather of equally shaped float64 rows through torch.distributed (nccl = RCCL on a
    GPU build, gloo on CPU).  With `device` the rows travel through a device tensor (pinned host
    staging and device buffers are allocated once per row shape).  The returned callable is
    blocking; its `.start(rows)` attribute launches the collective asynchronously and returns an
    object whose `.wait()` yields the gathered array."""
    import torch
    import torch.distributed as dist
    bufs = {}

    def start(rows: np.ndarray) -> _PendingGather:
        world = dist.get_world_size()
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        if not device:
            t = torch.from_numpy(rows)
            out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype)
            return _PendingGather(dist.all_gather_into_tensor(out, t, async_op=True), out)
        key = rows.shape
        if key not in bufs:      # two sets: a gather may still be in flight when the next one starts
            bufs[key] = [dict(h_in=torch.empty(key, dtype=torch.float64).pin_memory(),
                              d_in=torch.empty(key, dtype=torch.float64, device=device),
                              d_out=torch.empty((world * key[0], key[1]), dtype=torch.float64, device=device),
                              h_out=torch.empty((world * key[0], key[1]), dtype=torch.float64).pin_memory())
                         for _ in range(2)] + [0]
        sets = bufs[key]
        b = sets[sets[2]]
        sets[2] ^= 1
        b["h_in"].numpy()[:] = rows
        b["d_in"].copy_(b["h_in"], non_blocking=True)
        return _PendingGather(dist.all_gather_into_tensor(b["d_out"], b["d_in"], async_op=True), b["d_out"], b["h_out"])

    def gather(rows: np.ndarray) -> np.ndarray:
        return start(rows).wait()

    gather.start = start
    return gather
