#!/usr/bin/env python3
"""Condenses tools/profile_marg.sh's rocprofv3 outputs into gpurun_out/<tag>_marg_summary.{md,json}: per-kernel times and the
raw PMC means per dispatch of the marginalised mode's kernels, with the derived fractions of k_star_marg."""
import csv, glob, json, collections, os, shutil, sys
tag = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from base_amd import build as _build
out = {"tag": tag, "csrc_sha256": _build.source_hash(), "command": "python3 tools/time_marg.py 50000 4 4 8 (under rocprofv3: --kernel-trace --stats, and separate --pmc passes)", "kernels": {}, "pmc": {}}
short = lambda n: n.split("(")[0].replace("void ", "")
for f in glob.glob(f"gpurun_out/{tag}_marg_trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"gpurun_out/{tag}_marg_kernel_stats.csv")
    for r in csv.DictReader(open(f)):
        out["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                            "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sca"):
    for f in glob.glob(f"gpurun_out/{tag}_marg_{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            out["pmc"].setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
for k, c in out["pmc"].items():         # MI355X_MICROARCH.md "HBM": FETCH_SIZE (KiB) reads 1/2 of the bytes on gfx950 -> doubled; WRITE_SIZE exact
    if "FETCH_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0
HBM_PEAK, CLOCK, N_SIMD, EVALS = 8.0e12, 2.4e9, 1024, 400000.0
_main = sorted((k for k in out["pmc"] if k.startswith("k_star_marg<") and k in out["kernels"]), key=lambda k: -out["kernels"][k]["calls"] * out["kernels"][k]["avg_us"])[:1]
for k, c in out["pmc"].items():
    if k not in _main:
        continue
    t = out["kernels"][k]["avg_us"] * 1e-6
    wc = max(1.0, c.get("SQ_WAVE_CYCLES", 1.0))
    out["roofline_k_star_marg"] = {
        "kernel": k, "avg_launch_us": out["kernels"][k]["avg_us"],
        "valu_issue_frac": 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (N_SIMD * CLOCK * t),
        "valu_insts_per_star_eval": c.get("SQ_INSTS_VALU", 0.0) / EVALS, "salu_insts_per_star_eval": c.get("SQ_INSTS_SALU", 0.0) / EVALS,
        "smem_insts_per_star_eval": c.get("SQ_INSTS_SMEM", 0.0) / EVALS, "vmem_rd_insts_per_star_eval": c.get("SQ_INSTS_VMEM_RD", 0.0) / EVALS,
        "hbm_bytes_per_launch": c.get("hbm_bytes_per_launch"), "hbm_frac": c.get("hbm_bytes_per_launch", 0.0) / t / HBM_PEAK,
        "l2_hit_rate": c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)),
        "wave_cycles": {"issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "waiting (s_waitcnt)": c.get("SQ_WAIT_ANY", 0.0) / wc,
                        "issue-stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / wc, "of which LDS": c.get("SQ_WAIT_INST_LDS", 0.0) / wc},
        "waves": c.get("SQ_WAVES")}
json.dump(out, open(f"gpurun_out/{tag}_marg_summary.json", "w"), indent=1)
with open(f"gpurun_out/{tag}_marg_summary.md", "w") as f:
    f.write(f"# marginalised mode under rocprofv3 ({tag}; kernel sources {out['csrc_sha256'][:16]})\n\n`{out['command']}`\n\n| kernel | calls | avg us | min | max | % |\n|---|---|---|---|---|---|\n")
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["pct"]):
        f.write(f"| `{k}` | {v['calls']} | {v['avg_us']:.1f} | {v['min_us']:.1f} | {v['max_us']:.1f} | {v['pct']:.1f} |\n")
    r = out.get("roofline_k_star_marg", {})
    f.write("\n## k_star_marg (per launch = 400 000 star-evals)\n\n```\n" + json.dumps(r, indent=1) + "\n```\n\nraw counters (mean per dispatch):\n\n```\n")
    f.write(json.dumps({k: v for k, v in out["pmc"].items() if k.startswith(("k_star_marg", "k_marg_table"))}, indent=1) + "\n```\n")
print(open(f"gpurun_out/{tag}_marg_summary.md").read()[:3000])
