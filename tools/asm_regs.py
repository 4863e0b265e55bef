"""Register / LDS / scratch usage per kernel from the .amdhsa metadata of a hipcc -S file."""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat not in name:
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk)
    vals = {k: int(g(k).group(1)) for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size") if g(k)}
    agpr = int(re.match(r"\s*(\d+)", blk).group(1))
    print(f"{name[:60]:60s} vgpr {vals.get('vgpr_count')} agpr {agpr} sgpr {vals.get('sgpr_count')} scratch {vals.get('private_segment_fixed_size')} lds {vals.get('group_segment_fixed_size')}")
