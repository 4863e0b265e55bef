"""Print the instruction-class sequence of a kernel (M=mfma v=valu t=trans d=ds-read D=ds-write g=vmem-load A=atomic/store w=waitcnt B=barrier s=salu |=label)."""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + pat + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
out = []
for l in lines[start:end]:
    if re.match(r"^\.LBB", l): out.append("|"); continue
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if not m: continue
    i = m.group(1)
    if i.startswith("v_mfma"): out.append("M")
    elif i.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq")): out.append("t")
    elif i.startswith("v_"): out.append("v")
    elif i.startswith("ds_read") or i.startswith("ds_bperm"): out.append("d")
    elif i.startswith("ds_"): out.append("D")
    elif i.startswith(("global_load", "buffer_load", "scratch_load")): out.append("g")
    elif i.startswith(("global_atomic", "global_store", "scratch_store")): out.append("A")
    elif i.startswith("s_waitcnt"): out.append("w")
    elif i.startswith("s_barrier"): out.append("B")
    elif i.startswith("s_"): out.append("s")
s = "".join(out)
s = re.sub(r"s{3,}", lambda m: "s%d" % len(m.group(0)), s)
print(len(s)); 
for i in range(0, len(s), 200): print(s[i:i+200])
