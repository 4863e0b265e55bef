"""Per-launch durations (ms) of the hot kernels from a rocprofv3 kernel trace, next to the bench's event timings."""
import csv, glob, json, os, sys
d, benchjson = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
b = json.load(open(benchjson))
w = b["warmup"]
print(f"# {os.path.basename(f)}; bench: steps {b['steps']}, warmup {w}; stage_ms (HIP events, timed steps only): {b['stage_ms']}")
for key in ("k_moments_x<16, false", "k_moments_x<8, false", "k_moments<", "k_solve<16, false", "k_solve<8, false", "k_solve<32, false", "k_grads_t<16, false, false", "k_grads_t<8, false, false", "k_grads_x", "k_grads<"):
    rows = [r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]
    if not rows:
        continue
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    # launch order: warmup, the timed steps, then the sustained leg (same step, seconds of it)
    per = 2 if key == "k_grads<" and b["config"]["n_h"] > 16 else 1       # N_h = 17..32: one launch per 16 columns of F
    timed = ms[w * per:(w + b["steps"]) * per]
    rest = ms[(w + b["steps"]) * per:]
    print(f"{key:22s} warmup+timed launches {[round(x, 3) for x in ms[:(w + b['steps']) * per]]}  mean of the timed ones "
          f"{sum(timed) / max(1, len(timed)):.3f} ms; {len(rest)} later launches (sustained leg) mean "
          f"{(sum(rest) / len(rest)) if rest else float('nan'):.3f} ms")
