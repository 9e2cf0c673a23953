#!/usr/bin/env python3
"""ISA check of the ASYNCHRONOUS loads of the marginalised kernels: the scalar row loads (SRow, b9_star_marg.hip.h) and the
packed-fp32 box test's LDS reads (box_bound32: ds_read_b128 from inline assembly, one wait for the batch -- the same hazard with
VGPR destinations); the L2 warm-up's global loads (L2Warm) likewise, until their s_waitcnt vmcnt(0).

SRow::load issues s_load_dwordx8 / x16 / x2 from inline assembly and defers the s_waitcnt to SRow::wait.  To the compiler
the destination SGPRs are defined at the load's ISSUE; nothing tells its register allocator or its waitcnt insertion that
the data arrive later, so a copy, a spill (v_writelane) or any other use of those SGPRs between the load and the wait would
read stale data -- silently, and only in some instances.  This script compiles the kernels' device code and, in every
k_star_marg* / k_marg_step* instance, propagates "these SGPRs may hold a load still in flight" from each inline-assembly
s_load along the control-flow graph until an s_waitcnt lgkmcnt(0) (a forward may-dataflow over the basic blocks: the row loop
carries a load across its back edge by design) and reports any instruction that names such a register.

    python tools/check_async_sloads.py [asm file]      exit code 1 on a finding   (tests/test_isa.py runs it)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from base_amd import build as b  # noqa: E402

ASM = os.path.join(ROOT, "build", "b9_kernels_device.s")


def device_asm(force=False):
    src = os.path.join(b.CSRC, "b9_kernels.hip")
    deps = [os.path.join(b.CSRC, f) for f in os.listdir(b.CSRC) if f.endswith((".h", ".hip"))]
    if not force and os.path.exists(ASM) and all(os.path.getmtime(d) <= os.path.getmtime(ASM) for d in deps):
        return ASM
    os.makedirs(os.path.dirname(ASM), exist_ok=True)
    subprocess.run([b.HIPCC] + b.HIP_FLAGS + ["--cuda-device-only", "-S", "-o", ASM, "-x", "hip", src], check=True, capture_output=True)
    return ASM


def sgprs(operand, bank="s"):
    """Register numbers of one bank ("s" / "v") an operand names: s12, s[12:27]."""
    out = set()
    for m in re.finditer(r"\b%s\[(\d+):(\d+)\]" % bank, operand):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\b%s(\d+)\b" % bank, operand):
        out.add(int(m.group(1)))
    return out


def regs(operand):
    """Registers an operand names, as ("s", n) / ("v", n)."""
    return {("s", n) for n in sgprs(operand, "s")} | {("v", n) for n in sgprs(operand, "v")}


def parse_kernels(path):
    """{kernel: [(line number, text, inside inline asm)]} of the marginalised kernels."""
    out, cur, in_asm = {}, None, False
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if re.search(r"k_star_marg|k_marg_step", m.group(1)) else None
            if cur:
                out[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";"):
            continue
        if t.startswith(".") and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        out[cur].append((ln, t.split(";")[0].strip(), in_asm))
    return out


def check_kernel(instrs):
    """Forward may-dataflow over the kernel's basic blocks: which SGPRs may hold an asynchronous load still in flight."""
    # basic blocks
    blocks, labels, cur = [], {}, []
    for ln, t, a in instrs:
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            if cur:
                blocks.append(cur)
            cur = []
            labels[m.group(1)] = len(blocks)
            continue
        cur.append((ln, t, a))
        op = t.split()[0]
        if op == "s_branch" or op.startswith("s_cbranch") or op == "s_endpgm" or op.startswith("s_setpc"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, blk in enumerate(blocks):
        s_ = []
        if blk:
            op = blk[-1][1].split()[0]
            tgt = blk[-1][1].split()[-1]
            if op == "s_branch":
                s_ = [labels[tgt]] if tgt in labels else []
            elif op.startswith("s_cbranch"):
                s_ = ([labels[tgt]] if tgt in labels else []) + ([i + 1] if i + 1 < len(blocks) else [])
            elif op == "s_endpgm" or op.startswith("s_setpc"):
                s_ = []
            else:
                s_ = [i + 1] if i + 1 < len(blocks) else []
        else:
            s_ = [i + 1] if i + 1 < len(blocks) else []
        succ.append(s_)
    state_in = [dict() for _ in blocks]
    vm = set()            # registers an inline-assembly global load defines (waited for by vmcnt, not lgkmcnt)
    findings, n_loads = {}, 0
    work = list(range(len(blocks)))
    counted = set()
    while work:
        i = work.pop()
        infl = dict(state_in[i])
        for ln, t, a in blocks[i]:
            op = t.split()[0]
            if op.startswith("s_waitcnt"):
                if "lgkmcnt(0)" in t:
                    infl = {r: l for r, l in infl.items() if r in vm}
                if "vmcnt(0)" in t:
                    infl = {r: l for r, l in infl.items() if r not in vm}
                continue
            if a and op.startswith("global_load"):          # (L2Warm: destinations pending until s_waitcnt vmcnt(0))
                if ln not in counted:
                    counted.add(ln)
                    n_loads += 1
                for r in regs(t.split(",")[0]):
                    infl[r] = ln
                    vm.add(r)
                continue
            if a and (op.startswith("s_load_dword") or op.startswith("ds_read")):
                if ln not in counted:
                    counted.add(ln)
                    n_loads += 1
                if op.startswith("ds_read"):          # (its address register is read at issue: only a destination may not be named)
                    hit = sorted(regs(t[len(op):]) & set(infl))
                    if hit:
                        findings[ln] = (ln, t, [b_ + str(n) for b_, n in hit], infl[hit[0]])
                for r in regs(t.split(",")[0]):
                    infl[r] = ln
                continue
            if not infl or op in ("s_endpgm", "s_branch", "s_barrier", "s_nop") or op.startswith("s_cbranch") or op.startswith("s_sleep"):
                continue
            hit = sorted(regs(t[len(op):]) & set(infl))
            if hit:
                findings[ln] = (ln, t, [b_ + str(n) for b_, n in hit], infl[hit[0]])
        for j in succ[i]:
            merged = dict(state_in[j])
            merged.update({r: l for r, l in infl.items() if r not in merged})
            if merged.keys() != state_in[j].keys():
                state_in[j] = merged
                work.append(j)
    return list(findings.values()), n_loads


def check(path):
    findings, n_loads, kernels = [], 0, parse_kernels(path)
    for name, instrs in kernels.items():
        f, n = check_kernel(instrs)
        n_loads += n
        findings += [(name,) + x for x in f]
    return findings, n_loads, len(kernels)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else device_asm()
    findings, n_loads, n_kernels = check(path)
    print(f"{n_kernels} marginalised kernel instances, {n_loads} asynchronous loads (scalar rows, LDS box words) followed to their waits: {len(findings)} finding(s)")
    for fn, ln, text, regs, at in findings[:40]:
        print(f"  {fn[:60]} line {ln}: `{text[:90]}` names {regs} while the load of line {at} is in flight")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
