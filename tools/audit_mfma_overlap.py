#!/usr/bin/env python3
"""Find the VALU -> MFMA hazard hipcc does not cover for inline-asm writers (gfx950, measured round 2).

An MFMA whose destination overlaps its A or B operand returns garbage when a VALU instruction wrote one of those
operand registers fewer than 2 wait states earlier (tools/ubench/mfma_overlap.hip: wrong with no nop and with
`s_nop 0`, right from `s_nop 1` on; without the overlap no wait is needed).  hipcc gives the destination of an MFMA
that starts from C = 0 the registers of an operand that dies there, and it puts only `s_nop 0` between the kernels'
inline-asm v_cvt_pk_bf16_f32 and such an MFMA: k_grads_s3 came back 2e-3 off.  The kernels keep such operands live
(qfa_common.h six_terms); this audit lists every overlapping MFMA whose operand is written within the two preceding
wait states, and the build fails on any.

    tools/audit_mfma_overlap.py file.s [...]         (-v also counts the harmless overlaps)
"""
import re
import sys

REG = re.compile(r"([va])\[(\d+):(\d+)\]|([va])(\d+)")
NEED = 2                                   # wait states between the VALU write and the MFMA


def span(tok):
    m = REG.fullmatch(tok.strip())
    if not m:
        return None
    if m.group(1):
        return m.group(1), int(m.group(2)), int(m.group(3))
    return m.group(4), int(m.group(5)), int(m.group(5))


def overlap(a, b):
    return bool(a and b and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2])


def written(ins):
    """register spans a vector instruction writes (first operand; both for the swaps)"""
    parts = ins.split(None, 1)
    if len(parts) < 2 or not parts[0].startswith("v_") or parts[0].startswith("v_cmp"):
        return []
    ops = [o.strip() for o in parts[1].split(",")]
    n = 2 if "swap" in parts[0] else 1
    return [s for s in (span(o) for o in ops[:n]) if s]


def main(argv):
    verbose = "-v" in argv
    paths = [a for a in argv if a != "-v"]
    bad = total = 0
    for p in paths:
        kern = "?"
        window = []                         # (instruction, wait states it takes) of the recent instructions
        for n, line in enumerate(open(p), 1):
            s = line.split(";")[0].strip()
            m = re.match(r"(_Z\w+):", s)
            if m:
                kern, window = m.group(1), []
                continue
            if not s or s.startswith(".") or s.endswith(":"):
                continue
            if s.startswith("v_mfma") or s.startswith("v_smfmac"):
                ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
                d = span(ops[0])
                srcs = [o for o in (span(ops[1]), span(ops[2])) if overlap(d, o)]
                if srcs:
                    total += 1
                    ws = 0
                    for ins, w in reversed(window):
                        if ws >= NEED:
                            break
                        if any(overlap(wr, o) for wr in written(ins) for o in srcs):
                            bad += 1
                            print(f"{p}:{n}: {kern[:40]}: `{ins}` {ws} wait state(s) before `{s}`")
                            break
                        ws += w
            w = 1
            m = re.match(r"s_nop (\d+)", s)
            if m:
                w = int(m.group(1)) + 1
            window.append((s, w))
            del window[:-8]
    if verbose:
        print(f"{total} MFMAs whose destination overlaps an operand")
    print(f"{bad} hazardous MFMAs (operand written < {NEED} wait states before an overlapping destination)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
