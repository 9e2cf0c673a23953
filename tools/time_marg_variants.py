#!/usr/bin/env python3
"""Times tools/time_marg.py's bench shape under several library variants (build/variants/lib_<name>.so)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:]:
    env = dict(os.environ)
    if name != "default":
        env["B9_HIP_LIB"] = os.path.join(root, "build", "variants", f"lib_{name}.so")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "time_marg.py"), "50000", "4", "4", "8"], env=env, capture_output=True, text=True)
    print(f"{name:12s} {r.stdout.strip()} {r.stderr.strip()[-300:]}")
