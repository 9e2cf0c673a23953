#!/usr/bin/env python3
"""Times the star-likelihood kernel (HIP events, via the C ABI) under different launch plans.
Usage on the GPU box:  python tools/tune_k1.py [n_stars] [n_walkers] [wd_frac]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from base_amd import abi, engine, synth

n_stars = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
n_walkers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wd_frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
pack_d = synth.make_pack("parsec", 8)
truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, n_stars, seed=9003, truth=truth, wd_frac=wd_frac)
pack, stars = abi.make_pack(pack_d), abi.make_stars(cl)
priors, options = synth.default_priors(pack_d, truth), abi.make_options()
params = synth.walker_params(truth, n_walkers, seed=42, scale=float(os.environ.get("TUNE_SCALE", "0.05")))
d_params = torch.tensor(params, device="cuda")
d_out = torch.empty(n_walkers, dtype=torch.float64, device="cuda")
plans = [("auto", None, None)] + ([] if os.environ.get("TUNE_AUTO_ONLY") else [(f"tpb{t}", t, None) for t in (1, 2, 4)])
for name, tpb, lds in plans:
    for k, v in (("B9_TILES_PER_BLOCK", tpb), ("B9_FORCE_LDS", lds)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    eng = engine.Engine(pack, stars, priors, options)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        eng.logpost_device(d_params.data_ptr(), n_walkers, d_out.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    eng.enable_timing(1); eng.kernel_time_ms(True)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.logpost_device(d_params.data_ptr(), n_walkers, d_out.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200
    ms, n = eng.kernel_time_ms(True)
    by = eng.bytes_per_star_eval() * n_stars * n_walkers
    print(f"{name:14s} k1 {1e3*ms/n:8.2f} us  ({by/(ms/n*1e-3)/1e9:8.1f} GB/s alg)  wall/call {1e6*wall:8.2f} us  lp0 {d_out[0].item():.6f}", flush=True)
    eng.close()
