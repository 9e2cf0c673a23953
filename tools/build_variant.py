#!/usr/bin/env python3
"""Build a diagnostic / ablation variant of libbase9hip.so into build/variants/lib_<name>.so (never shipped; select it
with B9_HIP_LIB=...).   usage: build_variant.py <name> [-DFLAG ...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from base_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(b.ROOT, "build", "variants")
os.makedirs(out_dir, exist_ok=True)
out = os.path.join(out_dir, f"lib_{name}.so")
srcs = [os.path.join(b.CSRC, f) for f in b.HIP_SOURCES]
subprocess.run([b.HIPCC] + b.HIP_FLAGS + flags + ["-shared", "-o", out, "-x", "hip"] + srcs, check=True)
print(out)
