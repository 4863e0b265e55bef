"""Every s_waitcnt with a vmcnt field inside the loops (>= MIN instructions) of one kernel: python tools/loop_waits.py FILE.s KERNEL_REGEX [MIN]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 200
start = next(i for i, l in enumerate(lines) if re.match(r'^' + sys.argv[2] + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] >= mn:
        a = labels[m.group(1)]
        w = [f"{k}:{body[k].strip()}" for k in range(a, i) if 'vmcnt' in body[k]]
        print(f"loop {a}-{i} ({i - a} lines):", w)
