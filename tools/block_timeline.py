#!/usr/bin/env python3
"""Timeline of the LAST sampler block of a traced run (rocprofv3 --kernel-trace csv): every kernel / copy of the block with
its start offset, duration and the gap to its predecessor.   usage: block_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f))]
rows.sort()
# the last k_mcmc_finish before the marginalised leg closes the timed block; walk back to the copy / continue before its first step
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_mcmc_finish")]
end = idx[-1]
start = end
while start > 0 and (rows[start - 1][2].startswith("k_mcmc_step") or rows[start - 1][2].startswith("k_derive_iso") or
                     rows[start - 1][2].startswith("k_mcmc_continue") or "copyBuffer" in rows[start - 1][2]) and rows[start][0] - rows[start - 1][1] < 200000:
    start -= 1
t0 = rows[start][0]
print(f"last block: {end - start + 1} dispatches (+ what follows), times in us from its first dispatch")
prev_end = None
n_step = 0
for s, e, name in rows[start:end + 3]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = e
    if name.startswith("k_mcmc_step"):
        n_step += 1
        if 2 < n_step:      # print only the first two steps in full
            last_step = (s, e, gap)
            continue
    print(f"  {(s - t0) / 1e3:9.2f}  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  {name[:50]}")
print(f"  ... {n_step} k_mcmc_step launches; last one ends at {(last_step[1] - t0) / 1e3:.2f}")
