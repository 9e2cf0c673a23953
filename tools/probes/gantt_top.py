import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, ctypes as C
src = open("tools/gantt_marg.py").read()
# run the tool's setup up to the buffer read, then analyse
pre = src.split("print(f\"{name} marginalised")[0]
sys.argv = ["gantt_marg.py", sys.argv[1]]
exec(compile(pre, "gantt_pre", "exec"))
s = steps[-2]
a, b, role = s[:, 0], s[:, 1], s[:, 2] & 0xFF
t0 = a.min()
idx = np.where(t[order[-2]][:, 1] > 0)[0]
d = (b - a) / 100.0
top = np.argsort(-d)[:12]
for k in top:
    print(f"wg {idx[k]:5d} role {role[k]} xcc {(s[k,2]>>8)&15} start {(a[k]-t0)/100:6.2f} dur {d[k]:6.2f} end {(b[k]-t0)/100:6.2f}")
m = role == 0
print("star wg durations by dispatch position (first 40):", " ".join(f"{x:.1f}" for x in d[m][:40]))
print("... last 20:", " ".join(f"{x:.1f}" for x in d[m][-20:]))
