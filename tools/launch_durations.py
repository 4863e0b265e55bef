"""Per-launch durations (ms) of the hot kernels from a rocprofv3 kernel trace, next to the bench's event timings."""
import csv, glob, json, os, sys
d, benchjson = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
b = json.load(open(benchjson))
w = b["warmup"]
print(f"# {os.path.basename(f)}; bench: steps {b['steps']}, warmup {w}; stage_ms (HIP events, timed steps only): {b['stage_ms']}")
for key in ("k_moments_x", "k_moments<", "k_solve", "k_grads"):
    rows = [r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]
    if not rows:
        continue
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    per = len(ms) // (b["steps"] + w)
    timed = ms[w * per:]
    print(f"{key:12s} launches {[round(x, 3) for x in ms]}  mean of the timed ones {sum(timed) / len(timed) * per:.3f} ms per step")
