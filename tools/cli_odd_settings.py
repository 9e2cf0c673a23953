import os, subprocess, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np
from base_amd import synth, abi
d = tempfile.mkdtemp(prefix="b9odd_")
pack_d = synth.make_pack("dsed", 5, n_feh=3, n_age=5, n_eep=40); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 77, seed=3, truth=truth, wd_frac=0.1)
root = synth.write_models_dir(pack_d, os.path.join(d, "models"))
phot = synth.write_phot(cl, pack_d["filters"], os.path.join(d, "c.phot"))
exe = "/root/repo/base_amd/host/bin/"
for args in (["--burnIter", "0", "--runIter", "10", "--thin", "7", "--walkers", "3", "--block", "1000"],
             ["--burnIter", "5", "--runIter", "1", "--walkers", "1", "--block", "1"],
             ["--burnIter", "33", "--runIter", "17", "--thin", "100", "--walkers", "2"]):
    yml = synth.write_yaml(os.path.join(d, "base9.yaml"), phot, root, os.path.join(d, "run"), truth, ms_model="dsed", burn=10, run=10, walkers=1)
    r = subprocess.run([exe + "singlePopMcmc", "--config", yml] + args, capture_output=True, text=True)
    n = sum(1 for _ in open(os.path.join(d, "run.res"))) - 2
    print(args, "rc", r.returncode, "rows", n, r.stderr.strip().split("\n")[-1][:150])
    r2 = subprocess.run([exe + "sampleMass", "--config", yml] + args[:0], capture_output=True, text=True)
    print("   sampleMass rc", r2.returncode, r2.stderr.strip().split("\n")[-1][:160])
r = subprocess.run([exe + "makeCMD", "--config", yml], capture_output=True, text=True); print("makeCMD rc", r.returncode)
