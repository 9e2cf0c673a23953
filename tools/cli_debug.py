"""Debug aid: run singlePopMcmc from a start AWAY from the truth and show where the chain goes.

  python tools/cli_debug.py [--pack dsed] [--stars 1500] [--burn 1500] [--run 1500] [--dfeh -0.02]
"""
import argparse, os, subprocess, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np
from base_amd import synth, abi

ap = argparse.ArgumentParser()
ap.add_argument("--pack", default="parsec"); ap.add_argument("--stars", type=int, default=5000)
ap.add_argument("--burn", type=int, default=3000); ap.add_argument("--run", type=int, default=2000)
ap.add_argument("--dage", type=float, default=0.01); ap.add_argument("--dmod", type=float, default=0.02)
ap.add_argument("--dfeh", type=float, default=0.0); ap.add_argument("--seed", type=int, default=3)
ap.add_argument("--wd-frac", type=float, default=0.0)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="b9dbg_")
pack_d = synth.make_pack(a.pack, 8, n_feh=4, n_age=8, n_eep=90); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, a.stars, seed=a.seed, truth=truth, wd_frac=a.wd_frac)
root = synth.write_models_dir(pack_d, os.path.join(d, "models"))
phot = synth.write_phot(cl, pack_d["filters"], os.path.join(d, "c.phot"))
start = truth.copy(); start[abi.P_LOGAGE] += a.dage; start[abi.P_MOD] += a.dmod; start[abi.P_FEH] += a.dfeh
yml = synth.write_yaml(os.path.join(d, "base9.yaml"), phot, root, os.path.join(d, "run"), start, ms_model=a.pack,
                       burn=a.burn, run=a.run, walkers=4)
exe = "/root/repo/base_amd/host/bin/singlePopMcmc"
r = subprocess.run([exe, "--config", yml, "--verbose", "--priorDistMod", repr(float(truth[abi.P_MOD])),
                    "--priorFe_H", repr(float(truth[abi.P_FEH])), "--priorAv", repr(float(truth[abi.P_ABS]))],
                   capture_output=True, text=True)
lines = r.stderr.strip().split("\n")
print("\n".join(lines[:12])); print("..."); print("\n".join(lines[-4:]))
res = np.loadtxt(os.path.join(d, "run.res"), skiprows=2)
main = res[res[:, -1] == 3]
print("truth ", truth[[0, 2, 3, 4]]); print("start ", start[[0, 2, 3, 4]])
print("mean  ", main[:, :4].mean(axis=0)); print("std   ", main[:, :4].std(axis=0))
n = len(res) // 8
for i in range(8):
    seg = res[i * n:(i + 1) * n]
    print(f"  eighth {i}: mean {seg[:, :4].mean(axis=0)}  logPost {seg[:, -2].mean():.2f}")
print("unique logAge values in main run:", len(np.unique(main[:, 0])))
