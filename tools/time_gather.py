#!/usr/bin/env python3
"""Host cost of one start()/wait() of the sampler's all-gather on a 1-rank RCCL group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from base_amd import mcmc
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
g = mcmc.torch_all_gather("cuda")
rows = np.random.default_rng(0).normal(size=(8, 39))
for _ in range(20): g(rows)
for label, fn in (("start+wait", lambda: g.start(rows).wait()),):
    t0 = time.perf_counter()
    for _ in range(200): fn()
    print(f"{label}: {1e6*(time.perf_counter()-t0)/200:.1f} us")
t0 = time.perf_counter()
for _ in range(200): p = g.start(rows)
torch.cuda.synchronize(); print(f"start only: {1e6*(time.perf_counter()-t0)/200:.1f} us")
t = torch.empty(8, 39, dtype=torch.float64, device="cuda"); o = torch.empty(8, 39, dtype=torch.float64, device="cuda")
t0 = time.perf_counter()
for _ in range(200): dist.all_gather_into_tensor(o, t)
torch.cuda.synchronize(); print(f"bare all_gather_into_tensor: {1e6*(time.perf_counter()-t0)/200:.1f} us")
t0 = time.perf_counter()
for _ in range(200): dist.barrier()
print(f"barrier: {1e6*(time.perf_counter()-t0)/200:.1f} us")
dist.destroy_process_group()
