"""Audit of the compiled k_grads_x (qfa_amd/csrc: `make asmgx` -> /tmp/qfa_gx.s): the spectra prefetch is issued by asm
statements that hipcc does not track, so nothing but program order protects their destination registers.  For every
such load (between ;;#ASMSTART / ;;#ASMEND) check that no instruction reads or writes its destination registers before
the next `s_waitcnt vmcnt(...)`; also require zero scratch (spills of loop-carried registers produced wrong results
in this kernel once).  Exit code 1 on a violation."""
import re, sys
src = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/qfa_gx.s").read()
bad = 0
for m in re.finditer(r"^(_Z9k_grads_x\w+):[^\n]*\n(.*?)^\.Lfunc_end", src, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    in_asm = False
    pending = {}                                  # register number -> line of the load
    nload = 0
    for ln, line in enumerate(body):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if "scratch_" in t:
            print(f"{name}: scratch access: {t}"); bad += 1
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
    print(f"{name}: {nload} asm loads audited")
sys.exit(1 if bad else 0)
