import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29514"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
if backend == "nccl": torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", 0))
elif backend == "gloo": torch.distributed.init_process_group("gloo")
from base_amd import abi, engine, mcmc, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
start = synth.walker_params(truth, 8, seed=42, scale=0.02)
gather = mcmc.torch_all_gather("cuda" if backend == "nccl" else None) if backend != "none" else None
s = mcmc.WalkerSampler(start, mcmc.DeviceBlockRunner(eng), 0, 1, gather, seed=2024, block=100)
s.initialise(eng.logpost)
T = {"collect": 0.0, "gather": 0.0, "consume": 0.0, "submit": 0.0}
oc, os_, osg, ofg = s.runner.collect, s.runner.submit, s._start_gather, s._finish_gather
def tc(h):
    t0 = time.perf_counter(); r = oc(h); T["collect"] += time.perf_counter() - t0; return r
def ts(*a, **k):
    t0 = time.perf_counter(); r = os_(*a, **k); T["submit"] += time.perf_counter() - t0; return r
def tsg(row):
    t0 = time.perf_counter(); r = osg(row); T["gather"] += time.perf_counter() - t0; return r
def tfg(p):
    t0 = time.perf_counter(); r = ofg(p); T["gather"] += time.perf_counter() - t0; return r
s.runner.collect, s.runner.submit, s._start_gather, s._finish_gather = tc, ts, tsg, tfg
s.run(500)
for k in T: T[k] = 0.0
t0 = time.perf_counter(); s.run(3000); dt = time.perf_counter() - t0
print(backend, "us/step %.2f" % (dt / 3000 * 1e6), {k: round(v / 30 * 1e6) for k, v in T.items()}, "us per block")
