"""Instruction mix of the innermost hot loop of a kernel (between 'Inner Loop Header' and the last backward branch)."""
import re, sys
from collections import Counter
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + pat + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
hdr = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
if not hdr:
    print("no loop"); sys.exit()
def mix(h, back):
    c = Counter()
    for l in body[h:back + 1]:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m: continue
        i = m.group(1)
        if i.startswith("v_mfma"): c["mfma"] += 1
        elif i.startswith("v_accvgpr"): c["accvgpr"] += 1
        elif i.startswith(("global_load", "buffer_load")): c["gload"] += 1
        elif i.startswith("global_atomic"): c["gatomic"] += 1
        elif i.startswith("ds_"): c["ds"] += 1
        elif i.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq")): c["trans"] += 1
        elif i.startswith("v_"): c["valu"] += 1
        elif i.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif i.startswith("scratch"): c["scratch"] += 1
        elif i.startswith("s_"): c["salu"] += 1
    return dict(sorted(c.items()))
for h in hdr:
    label = body[h].split(":")[0].strip()
    backs = [i for i, l in enumerate(body) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\s*$", l)]
    if backs and max(backs) - h > 100:
        print(pat, label, "lines", max(backs) - h, mix(h, max(backs)))
