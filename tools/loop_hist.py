"""Instruction histogram of every loop (backward branch) of one kernel in a -save-temps .s file.
usage: python tools/loop_hist.py FILE.s KERNEL_REGEX"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^' + sys.argv[2] + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
print('kernel lines', len(body))
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        c = Counter(re.match(r'^\s+([a-z_0-9]+)', x).group(1) for x in body[a:i] if re.match(r'^\s+([a-z_0-9]+)', x))
        tot = sum(c.values())
        if tot < 40: continue
        valu = sum(n for k, n in c.items() if k.startswith('v_') and 'mfma' not in k)
        print(f"loop {a}-{i}: {tot} instrs, mfma {sum(n for k, n in c.items() if 'mfma' in k)}, valu {valu} (v_mov {c['v_mov_b32_e32'] + c['v_mov_b64_e32']}, pk {sum(n for k, n in c.items() if k.startswith('v_pk_'))}, trans {sum(n for k, n in c.items() if k.startswith(('v_exp', 'v_log', 'v_rcp')))}), "
              f"saveexec {sum(n for k, n in c.items() if 'saveexec' in k)}, ds {sum(n for k, n in c.items() if k.startswith('ds_'))}, salu {sum(n for k, n in c.items() if k.startswith('s_') and not k.startswith(('s_waitcnt', 's_nop')))}, waitcnt {c['s_waitcnt']}, nop {c['s_nop']}, vmem {sum(n for k, n in c.items() if k.startswith(('global_', 'scratch_')))}")
