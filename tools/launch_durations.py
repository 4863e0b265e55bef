"""Per-launch durations (ms) of the hot kernels from a rocprofv3 kernel trace, next to the bench's event timings.
(Round 5: the legs of one bench run launch different instantiations -- the headline, sustained and factored legs the factored-z
ones, the first sighting of a tensor and the `zabs_kernels` leg the zabs ones -- so every instantiation is listed on its own:
launches, mean, min, max, the first launches in order.)"""
import csv, glob, json, os, re, sys
d, benchjson = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
b = json.load(open(benchjson))
print(f"# {os.path.basename(f)}; bench: steps {b['steps']}, warmup {b['warmup']}; stage_ms (HIP events, timed steps only): {b['stage_ms']}")
by = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not re.search(r"\bk_(moments|solve|grads|s12|predict|zfactor|prep_step|finalize)", n):
        continue
    n = re.sub(r"^void ", "", n).split("(")[0]
    by.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for n in sorted(by):
    ms = by[n]
    print(f"{n:58s} {len(ms):5d} launches  mean {sum(ms) / len(ms):8.4f} ms  min {min(ms):8.4f}  max {max(ms):8.4f}  first {[round(x, 3) for x in ms[:8]]}")
