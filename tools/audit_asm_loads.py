"""Audit of the compiled kernels (qfa_amd/csrc/build/*-gfx950.s, written by the compile that produced the shipped objects: -save-temps): register prefetches are issued by
asm statements that hipcc does not track, so nothing but program order protects their destination registers.  For every
such load (between ;;#ASMSTART / ;;#ASMEND) of every kernel check that no instruction reads or writes its destination
registers before the next `s_waitcnt vmcnt(...)`; in k_grads_x and k_predict_x also require zero scratch (spills of
loop-carried registers produced wrong results in k_grads_x once, and a scratch reload makes hipcc wait vmcnt(0) in the
middle of the counted queue); in k_grads_t no instruction outside the asm statements may touch M0 (a stage's LDS-DMA
pieces share one write of it).  The scan is linear in program order (the code behind an unconditional branch starts
with nothing pending; a counted wait is taken to retire every asm load before it): a build-time tripwire beside the
dynamic check, tests/test_tracked_loads.py.  Exit code 1 on a violation.

    tools/audit_asm_loads.py [file.s ...]          (run by `make`: the library does not link unless it passes)"""
import re, sys
paths = sys.argv[1:] or ["qfa_amd/csrc/build/qfa_gx-hip-amdgcn-amd-amdhsa-gfx950.s"]
src = "\n".join(open(p).read() for p in paths)
NO_SCRATCH = ("_Z9k_grads_x", "_Z9k_grads_t", "_Z11k_predict_x")
bad = 0
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", src, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    in_asm = False
    pending = {}                                  # register number -> line of the load
    fallthrough = True
    nload = 0
    for ln, line in enumerate(body):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        if re.match(r"\.LBB\w+:", t):               # a label behind an unconditional branch: another path starts here
            if not fallthrough:
                pending = {}
            fallthrough = True
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.startswith("s_branch"):
            fallthrough = False
            continue
        if "scratch_" in t and name.startswith(NO_SCRATCH):
            print(f"{name}: scratch access: {t}"); bad += 1
        # k_grads_t (round 5) writes M0 once per stage and lets the stage's LDS-DMA pieces ride on it: no compiler-made
        # instruction may touch M0 in that kernel
        if not in_asm and name.startswith("_Z9k_grads_t") and re.search(r"\bm0\b", t):
            print(f"{name}: M0 used outside the asm statements: {t}"); bad += 1
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            pending.clear(); continue
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(x) for x in re.findall(r"\bv(\d+)\b", t))
        if in_asm and t.startswith("global_load_") and "lds" not in t:
            dst = re.match(r"global_load_\w+\s+(v\[(\d+):(\d+)\]|v(\d+))", t)
            d = set(range(int(dst.group(2)), int(dst.group(3)) + 1)) if dst.group(2) else {int(dst.group(4))}
            for r in d:
                pending[r] = ln
            nload += 1
            regs -= d
        hit = regs & set(pending)
        if hit:
            print(f"{name}: line {ln}: `{t}` touches {sorted(hit)} loaded at lines {sorted(set(pending[r] for r in hit))} before a vmcnt wait")
            bad += 1
    if nload or name.startswith(NO_SCRATCH):
        print(f"{name[:60]}: {nload} asm loads audited")
sys.exit(1 if bad else 0)
