"""Instruction mix per kernel from the hipcc -S output (tools; not product code)."""
import re, sys
from collections import Counter
path = sys.argv[1] if len(sys.argv) > 1 else "qfa_amd/csrc/build/qfa_capi-hip-amdgcn-amd-amdhsa-gfx950.s"
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
name = None
c = Counter()
def flush():
    if name and pat in name:
        print(name[:60], dict(sorted(c.items())))
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush(); name = m.group(1); c = Counter(); continue
    m = re.match(r"^\s+([a-z_0-9]+)", line)
    if not m or name is None: continue
    i = m.group(1)
    if i.startswith("v_mfma"): c["mfma"] += 1
    elif i.startswith(("global_load", "buffer_load")): c[i] += 1
    elif i.startswith("global_atomic"): c["atomic"] += 1
    elif i.startswith("global_store"): c["gstore"] += 1
    elif i.startswith("ds_"): c["ds"] += 1
    elif i.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq")): c["trans"] += 1
    elif i.startswith("v_"): c["valu"] += 1
    elif i.startswith("s_waitcnt"): c["waitcnt"] += 1
    elif i.startswith("scratch"): c["scratch"] += 1
    elif i.startswith("s_"): c["salu"] += 1
flush()
