"""ISA-level gate on the asynchronous loads of the marginalised kernels -- the scalar row loads and the packed-fp32 box
test's batched LDS reads (the SGPRs an inline-assembly
s_load defines are not really defined until the deferred s_waitcnt -- nothing stops a future compiler from copying or
spilling them in between).  tools/check_async_sloads.py follows every such load along the control-flow graph of the
generated ISA; here: the checker finds a planted violation, and the shipped kernels have none.  CPU only (hipcc
cross-compiles the device code; the assembly is cached under build/)."""
import importlib.util
import os
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_async_sloads", os.path.join(ROOT, "tools", "check_async_sloads.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def _asm(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text("_Z11k_star_margILi8EEv:\n" + textwrap.dedent(body) + "\n.Lfunc_end0:\n")
    return str(p)


def test_checker_sees_a_use_before_the_wait_also_across_a_back_edge(tmp_path):
    clean = """
        ;;#ASMSTART
        s_load_dwordx16 s[16:31], s[0:1], 0x0
        ;;#ASMEND
        v_fma_f64 v[0:1], v[2:3], s[40:41], v[4:5]
        ;;#ASMSTART
        s_waitcnt lgkmcnt(0)
        ;;#ASMEND
        v_fma_f64 v[0:1], v[2:3], s[16:17], v[4:5]
        s_endpgm
    """
    f, n, k = chk.check(_asm(tmp_path, clean))
    assert (len(f), n, k) == (0, 1, 1)
    spilled = clean.replace("v_fma_f64 v[0:1], v[2:3], s[40:41], v[4:5]", "v_writelane_b32 v9, s20, 3")
    f, n, k = chk.check(_asm(tmp_path, spilled))
    assert len(f) == 1 and f[0][3] == ["s20"]
    # the load issued at a loop's end is waited for at its head: a use on the path around the back edge is a finding,
    # the same registers used after the wait are not
    loop = """
        .LBB0_1:
        ;;#ASMSTART
        s_waitcnt lgkmcnt(0)
        ;;#ASMEND
        v_fma_f64 v[0:1], v[2:3], s[16:17], v[4:5]
        ;;#ASMSTART
        s_load_dwordx16 s[16:31], s[0:1], 0x0
        ;;#ASMEND
        s_cbranch_scc1 .LBB0_1
        s_mov_b32 s50, s31
        s_endpgm
    """
    f, n, k = chk.check(_asm(tmp_path, loop))
    assert len(f) == 1 and f[0][3] == ["s31"] and "s_mov_b32" in f[0][2]
    # the box test's LDS reads: two requests, one wait; a VALU instruction that names a destination in between is a finding
    lds = """
        ;;#ASMSTART
        ds_read_b128 v[60:63], v9
        ;;#ASMEND
        ;;#ASMSTART
        ds_read_b128 v[64:67], v9 offset:1024
        ;;#ASMEND
        v_mov_b32_e32 v70, v8
        ;;#ASMSTART
        s_waitcnt lgkmcnt(0)
        ;;#ASMEND
        v_pk_fma_f32 v[0:1], v[60:61], s[12:13], v[62:63]
        s_endpgm
    """
    f, n, k = chk.check(_asm(tmp_path, lds))
    assert (len(f), n, k) == (0, 2, 1)
    f, n, k = chk.check(_asm(tmp_path, lds.replace("v_mov_b32_e32 v70, v8", "v_mov_b32_e32 v70, v65")))
    assert len(f) == 1 and f[0][3] == ["v65"]
    # the L2 warm-up's global loads are waited for by vmcnt: an lgkmcnt wait does not release their registers
    warm = """
        ;;#ASMSTART
        global_load_dword v40, v[2:3], off
        ;;#ASMEND
        s_waitcnt lgkmcnt(0)
        v_add_u32_e32 v5, v6, v7
        ;;#ASMSTART
        s_waitcnt vmcnt(0)
        ;;#ASMEND
        v_mov_b32_e32 v41, v40
        s_endpgm
    """
    f, n, k = chk.check(_asm(tmp_path, warm))
    assert (len(f), n, k) == (0, 1, 1)
    f, n, k = chk.check(_asm(tmp_path, warm.replace("v_add_u32_e32 v5, v6, v7", "v_add_u32_e32 v40, v6, v7")))
    assert len(f) == 1 and f[0][3] == ["v40"]


def test_shipped_marginalised_kernels_touch_no_row_register_before_its_wait():
    findings, n_loads, n_kernels = chk.check(chk.device_asm())
    assert n_kernels >= 30 and n_loads >= 300, (n_kernels, n_loads)          # every NFP x populations x SAMPLE / SPLIT / COST instance, and k_marg_step's
    assert not findings, findings[:5]
