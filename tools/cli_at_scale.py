#!/usr/bin/env python3
"""Writes a 50k-star synthetic cluster + PARSEC-shaped model directory and runs the C++ singlePopMcmc on it."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from base_amd import synth
d = tempfile.mkdtemp(prefix="b9cli_")
pack_d = synth.make_pack("parsec", 8); truth = synth.default_params(pack_d)
cl = synth.make_cluster(pack_d, 50000, seed=9003, truth=truth)
t0 = time.time()
root = synth.write_models_dir(pack_d, os.path.join(d, "models"))
phot = synth.write_phot(cl, pack_d["filters"], os.path.join(d, "c.phot"))
yml = synth.write_yaml(os.path.join(d, "base9.yaml"), phot, root, os.path.join(d, "run"), truth, burn=2000, run=4000, walkers=8, thin=10)
print(f"inputs written in {time.time()-t0:.1f} s ({os.path.getsize(os.path.join(root, 'msrgb', 'parsec.model'))/1e6:.0f} MB model file)")
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "base_amd", "host", "bin", "singlePopMcmc")
t0 = time.time()
r = subprocess.run([exe, "--config", yml, "--block", "100"], capture_output=True, text=True)
print(r.stderr.strip()); print(f"wall (incl. parsing {os.path.getsize(phot)/1e6:.0f} MB phot + model files, staging): {time.time()-t0:.1f} s")
