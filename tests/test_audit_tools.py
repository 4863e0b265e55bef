"""The two build-time audits of the generated assembly (tools/audit_mfma_overlap.py, tools/audit_asm_loads.py; run by
`make` -- the library does not link unless both pass -- and so by __graft_entry__.build()) on hand-written snippets: each must flag the pattern it exists for and pass
the harmless neighbours."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tool, text, tmp_path):
    p = tmp_path / "k.s"
    p.write_text(text)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", tool), str(p)], capture_output=True, text=True)
    return r.returncode, r.stdout


KERNEL = "_Z6kernelv:\n{}\n\ts_endpgm\n.Lfunc_end0:\n"


def test_mfma_overlap_hazard_is_flagged(tmp_path):
    # the sequence hipcc emitted in k_grads_s3: conversion into v133, one wait state, MFMA v[130:133] <- v[132:133] x ...
    bad = KERNEL.format("\t;;#ASMSTART\n\tv_cvt_pk_bf16_f32 v133, v130, v131\n\t;;#ASMEND\n\ts_nop 0\n"
                        "\tv_mfma_f32_16x16x16_bf16 v[130:133], v[132:133], v[144:145], 0")
    rc, out = run("audit_mfma_overlap.py", bad, tmp_path)
    assert rc == 1 and "1 hazardous" in out
    # two wait states are enough (tools/ubench/mfma_overlap.hip)
    ok = bad.replace("s_nop 0", "s_nop 1")
    assert run("audit_mfma_overlap.py", ok, tmp_path)[0] == 0
    # no overlap: no wait needed
    sep = bad.replace("v[130:133], v[132:133]", "v[140:143], v[132:133]")
    assert run("audit_mfma_overlap.py", sep, tmp_path)[0] == 0
    # an overlap whose operand was written long before is harmless (hipcc allocates hundreds)
    old = KERNEL.format("\tv_cvt_pk_bf16_f32 v133, v130, v131\n\tv_add_f32_e32 v1, v2, v3\n\tv_add_f32_e32 v1, v2, v3\n"
                        "\tv_mfma_f32_16x16x16_bf16 v[130:133], v[132:133], v[144:145], 0")
    assert run("audit_mfma_overlap.py", old, tmp_path)[0] == 0
    # the B operand counts as well
    badb = KERNEL.format("\tv_perm_b32 v20, v1, v2, v3\n\tv_mfma_f32_16x16x32_bf16 v[20:23], v[4:7], v[20:23], v[8:11]")
    assert run("audit_mfma_overlap.py", badb, tmp_path)[0] == 1


def test_asm_load_destination_touched_before_the_wait_is_flagged(tmp_path):
    load = "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[20:23], v122, s[8:9] offset:0\n\t;;#ASMEND\n"
    bad = KERNEL.format(load + "\tv_add_f32_e32 v1, v21, v3\n\ts_waitcnt vmcnt(0)")
    rc, out = run("audit_asm_loads.py", bad, tmp_path)
    assert rc == 1 and "touches [21]" in out
    ok = KERNEL.format(load + "\tv_add_f32_e32 v1, v2, v3\n\ts_waitcnt vmcnt(0)\n\tv_add_f32_e32 v1, v21, v3")
    assert run("audit_asm_loads.py", ok, tmp_path)[0] == 0
    # the other arm of a branch (reached without the load) may use the same registers
    arm = KERNEL.format("\ts_cbranch_scc1 .LBB0_2\n" + load + "\ts_branch .LBB0_3\n.LBB0_2:\n\tv_mov_b32_e32 v20, v9\n"
                        ".LBB0_3:\n\ts_waitcnt vmcnt(0)")
    assert run("audit_asm_loads.py", arm, tmp_path)[0] == 0
    # LDS-DMA has no destination registers
    dma = KERNEL.format("\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v100, s[18:19]\n\t;;#ASMEND\n\tv_add_f32_e32 v1, v100, v3")
    assert run("audit_asm_loads.py", dma, tmp_path)[0] == 0


def test_scratch_is_refused_in_the_counted_kernels_only(tmp_path):
    body = "\tscratch_load_dword v4, off, off\n\ts_waitcnt vmcnt(0)"
    gx = "_Z9k_grads_xILi16ELb0EEvv:\n" + body + "\n\ts_endpgm\n.Lfunc_end0:\n"
    assert run("audit_asm_loads.py", gx, tmp_path)[0] == 1
    assert run("audit_asm_loads.py", KERNEL.format(body), tmp_path)[0] == 0
