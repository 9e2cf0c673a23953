#!/usr/bin/env python3
"""HIP API calls and GPU activity of the LAST sampler block of a run traced with
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace (csv).   usage: block_api_timeline.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
def load(pat):
    fs = sorted(glob.glob(d + "/**/*" + pat, recursive=True))
    return list(csv.DictReader(open(fs[0]))) if fs else []
k = load("kernel_trace.csv"); m = load("memory_copy_trace.csv"); a = load("hip_api_trace.csv")
ev = []
for r in k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU  " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]))
for r in m: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", r.get("Bytes", ""))))
for r in a: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "API  " + r["Function"]))
ev.sort()
fin = [i for i, e in enumerate(ev) if e[2].startswith("GPU  k_mcmc_finish")][-1]
t_end = ev[fin][1] + 200000
# the block's first GPU dispatch: walk back over k_mcmc_step kernels
gk = [e for e in ev if e[2].startswith("GPU") and e[0] <= ev[fin][0]]
i = len(gk) - 1
while i > 0 and gk[i][0] - gk[i - 1][1] < 100000: i -= 1
t0 = gk[i][0] - 300000
n_step_gpu = n_launch_api = 0
for s, e, name in ev:
    if s < t0 or s > t_end: continue
    if name.startswith("GPU  k_mcmc_step"):
        n_step_gpu += 1
        if 2 < n_step_gpu < 19: continue
    if name in ("API  hipLaunchKernel", "API  hipModuleLaunchKernel", "API  hipExtModuleLaunchKernel"):
        n_launch_api += 1
        if 3 < n_launch_api < 21: continue
    if name.startswith("API  hipEventCreate") or name.startswith("API  hipGetLastError") or name.startswith("API  __hipPushCallConfiguration") or name.startswith("API  __hipPopCallConfiguration"): continue
    print(f"{(s - t0) / 1e3:9.2f} +{(e - s) / 1e3:8.2f}  {name}")
