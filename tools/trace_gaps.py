#!/usr/bin/env python3
"""Per-kernel durations AND the gaps between consecutive kernels from a rocprofv3 --kernel-trace csv.
usage: trace_gaps.py <dir with *_kernel_trace.csv> [kernel-name substring to focus on]"""
import csv, glob, sys, collections
d = sys.argv[1]; focus = sys.argv[2] if len(sys.argv) > 2 else "k_"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev is not None:
        gap[(prev[0], name)].append(s - prev[1])
    prev = (name, e)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if focus in k:
        v2 = sorted(v); print(f"{k[:60]:60s} n={len(v):5d} mean {sum(v)/len(v)/1e3:8.2f} us  median {v2[len(v2)//2]/1e3:8.2f}  min {v2[0]/1e3:8.2f}")
print("gaps (end of A -> start of B):")
for k, v in sorted(gap.items(), key=lambda kv: -len(kv[1]))[:8]:
    v2 = sorted(v); print(f"  {k[0][:28]:28s} -> {k[1][:28]:28s} n={len(v):5d} median {v2[len(v2)//2]/1e3:7.2f} us  mean {sum(v)/len(v)/1e3:7.2f}")
