#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile_round.sh into gpurun_out/<tag>_summary.{md,json}."""
import csv, glob, json, collections, sys
tag = sys.argv[1]
out = {"tag": tag, "kernels": {}, "pmc": {}}
for f in glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].replace("void ", "")
        out["kernels"][name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                "pct": float(r["Percentage"])}
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(f"gpurun_out/{tag}_{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            out["pmc"].setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
for line in open(f"gpurun_out/{tag}_trace.log"):
    if line.startswith("{"):
        out["bench_line_under_profiler"] = json.loads(line)
# HBM traffic per launch of the dominant kernel: MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) reads exactly
# 1/2 of a coalesced stream's bytes on gfx950 -> doubled; WRITE_SIZE is exact.
for k, c in out["pmc"].items():
    if "FETCH_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0
json.dump(out, open(f"gpurun_out/{tag}_summary.json", "w"), indent=1)
with open(f"gpurun_out/{tag}_summary.md", "w") as md:
    md.write(f"# rocprofv3 summary {tag}\n\ncommand: `python3 bench.py --steps 400 --warmup 100 --no-cpu-baseline` under "
             "`rocprofv3 --kernel-trace --stats` and three separate `--pmc` passes\n\n")
    md.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["pct"]):
        md.write(f"| `{k}` | {v['calls']} | {v['avg_us']:.2f} | {v['min_us']:.2f} | {v['max_us']:.2f} | {v['pct']:.1f} |\n")
    md.write("\n## PMC (mean per dispatch)\n\n")
    for k, c in out["pmc"].items():
        if "copyBuffer" in k: continue
        md.write(f"* `{k}`: " + ", ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())) + "\n")
    md.write("\nFETCH_SIZE/WRITE_SIZE are KiB; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
             "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md section HBM).\n")
print(open(f"gpurun_out/{tag}_summary.md").read())
