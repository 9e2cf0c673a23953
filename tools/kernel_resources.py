#!/usr/bin/env python3
"""Compile b9_kernels.hip (device side only) with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs, AGPRs, scratch bytes per lane, occupancy, LDS, and the number of scratch_load / scratch_store instructions in its ISA
(a kernel whose SGPRs overflow into VGPR lanes reserves 4 B of stack per such VGPR that no instruction ever touches:
"scratch 20, instrs 0" is that, not a spill to memory).  Extra arguments are passed to hipcc (e.g. -DB9_K1_MIN_WAVES=4).

    python tools/kernel_resources.py [filter-substring] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from base_amd import build as b  # noqa: E402


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    flt = args[0] if args else ""
    src = os.path.join(b.CSRC, "b9_kernels.hip")
    asm = "/tmp/b9_kernel_resources.s"
    cmd = [b.HIPCC] + b.HIP_FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-S", "-o", asm, "-x", "hip", src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        raise SystemExit(1)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(\w[\w ]*): (.*?) \[-Rpass", line) or re.search(r"remark: +(\w[\w ]*): (.*?) \[-Rpass", line)
        if not m:
            m = re.search(r":\d+:\d+: remark: +(.*?): (.*?) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    n_scr, cur_fn = {}, None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur_fn = m.group(1)
        elif cur_fn and ("scratch_load" in line or "scratch_store" in line):
            n_scr[cur_fn] = n_scr.get(cur_fn, 0) + 1
    dem = subprocess.run(["c++filt"] + [r_["name"] for r_ in rows], capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s} {'scr.instrs':>10s}")
    for r_, d in zip(rows, dem):
        short = re.sub(r"\(.*", "", d).replace("void ", "")
        if flt and flt not in short:
            continue
        print(f"{short:58s} {r_.get('VGPRs', '?'):>5s} {r_.get('AGPRs', '?'):>5s} {r_.get('TotalSGPRs', '?'):>5s} "
              f"{r_.get('ScratchSize [bytes/lane]', '?'):>8s} {r_.get('Occupancy [waves/SIMD]', '?'):>4s} {r_.get('LDS Size [bytes/block]', '?'):>7s} {n_scr.get(r_['name'], 0):>10d}")


if __name__ == "__main__":
    main()
