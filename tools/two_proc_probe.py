#!/usr/bin/env python3
"""Upper bound for in-process chain pipelining: N processes x W walkers sharing one GPU."""
import os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = r'''
import os, sys, time
sys.path.insert(0, os.path.dirname(%r))
import numpy as np
from base_amd import abi, engine, mcmc, synth
W = int(sys.argv[1]); steps = int(sys.argv[2])
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
free = np.array(mcmc.DEFAULT_FREE); chol = np.diag([1e-5, 2e-5, 1e-5, 1e-5])
start = synth.walker_params(truth, W, seed=42, scale=0.02); lp = eng.logpost(start)
eng.mcmc_run_block(start, lp, np.arange(W), free, chol, 1, 0, 1500)
open(sys.argv[3], "w").write("ready"); 
while not os.path.exists(sys.argv[4]): time.sleep(0.001)
t0 = time.perf_counter(); eng.mcmc_run_block(start, lp, np.arange(W), free, chol, 1, 0, steps); dt = time.perf_counter() - t0
print("%%.2f" %% (1e6 * dt / steps))
''' % HERE
for nproc, W in ((1, 8), (2, 4), (2, 8), (4, 2)):
    go = f"/tmp/go_{nproc}_{W}"
    if os.path.exists(go): os.remove(go)
    procs, flags = [], []
    for p in range(nproc):
        flag = f"/tmp/ready_{nproc}_{W}_{p}"
        if os.path.exists(flag): os.remove(flag)
        flags.append(flag)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, str(W), "3000", flag, go], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    while not all(os.path.exists(f) for f in flags): time.sleep(0.01)
    open(go, "w").write("go")
    us = [float(p.communicate()[0].strip().split()[-1]) for p in procs]
    tot = sum(50000 * W / (u * 1e-6) for u in us)
    print(f"{nproc} process(es) x {W} walkers: us/step {us} -> {tot:.3e} star-evals/s total", flush=True)
