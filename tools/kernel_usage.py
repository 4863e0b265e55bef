"""Registers, scratch and LDS of every kernel of a build (the -save-temps assembly under qfa_amd/csrc/build).

    python tools/kernel_usage.py [pattern] [file.s ...]      (default: all four translation units)
"""
import glob
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
pat = args[0] if args and not args[0].endswith(".s") else ""
files = [a for a in args if a.endswith(".s")] or sorted(glob.glob(os.path.join(REPO, "qfa_amd/csrc/build/*gfx950.s")))
for f in files:
    src = open(f).read()
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?"
                         r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", src, re.S):
        ag, lds, name, scr, sg, vg = m.groups()
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"\(.*", "", d).replace("void ", "")
        if pat in d:
            print(f"{os.path.basename(f)[:8]:8s} {d:52s} vgpr {vg:>4s} agpr {ag:>4s} sgpr {sg:>4s} lds {lds:>6s} scratch {scr}")
