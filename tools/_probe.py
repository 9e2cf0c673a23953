import faulthandler, os, sys, time
faulthandler.dump_traceback_later(25, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from base_amd import abi, engine, hostlib, mcmc, synth
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
eng = engine.Engine(abi.make_pack(pack_d), abi.make_stars(cl), synth.default_priors(pack_d, truth), abi.make_options())
start = synth.walker_params(truth, 8, seed=42, scale=0.02)
lp = eng.logpost(start)
free = np.array(mcmc.DEFAULT_FREE, dtype=np.int32); chol = np.diag([mcmc.DEFAULT_STEP[int(k)] for k in free]) * 0.3
ids = np.arange(8, dtype=np.int32)
if "sync" in sys.argv:
    eng.mcmc_run_block(start, lp, ids, free, chol, 1, 0, 30, record=False); print("sync block ok", flush=True)
s = hostlib.HostSampler(8, free, [mcmc.DEFAULT_STEP[int(k)] for k in free], hostlib.Exchange.local(), seed=5, block=100, engine=eng)
print("created", flush=True)
s.initialise(start); print("initialised", flush=True)
os.environ["B9_SAMPLER_TRACE"] = "1"
s.run(300); print("run 300 ok", flush=True)
for n in (20, 100):
    t0 = time.perf_counter(); s.run(n); print(n, time.perf_counter() - t0, flush=True)
