#!/bin/bash
# GPU box: step legs and epoch leg of the given configs, one line each.  usage: tools/bench_epoch.sh c3 desi c1b ...
mkdir -p gpurun_out
for c in "$@"; do
  timeout -k 10 280 python bench.py --config $c --no-cpu-baseline --no-predict --sustain 0 ${BENCH_ARGS} > gpurun_out/be_$c.json 2> gpurun_out/be_$c.err || { echo "$c failed"; tail -5 gpurun_out/be_$c.err; continue; }
  python - <<PY
import json
d = json.load(open("gpurun_out/be_$c.json"))
f = d.get("factored_z") or {"ms_per_step": 0, "stage_ms": {"pass1_moments": 0, "pass2_grads": 0}}
e = d.get("epoch") or {}
i = e.get("indexed_step") or {"ms_per_step": 0, "stage_ms": {"pass1_moments": 0, "pass2_grads": 0}}
j = i.get("rows_in_storage_order") or i
print("      indexed step %.3f (p1 %.3f p2 %.3f); rows in storage order %.3f (p1 %.3f p2 %.3f)" % (i["ms_per_step"], i["stage_ms"]["pass1_moments"], i["stage_ms"]["pass2_grads"], j["ms_per_step"], j["stage_ms"]["pass1_moments"], j["stage_ms"]["pass2_grads"]))
print("%-5s zabs step %.3f (p1 %.3f p2 %.3f) | zfac step %.3f (p1 %.3f p2 %.3f) | epoch %.3f ms/step = %.3f of zfac step, %.3f of headline; graph %s, %s batches x %s" % (
    "$c", d["ms_per_step"], d["stage_ms"]["pass1_moments"], d["stage_ms"]["pass2_grads"], f["ms_per_step"], f["stage_ms"]["pass1_moments"],
    f["stage_ms"]["pass2_grads"], e.get("ms_per_step", 0), e.get("vs_step_only_factored_z", 0), e.get("vs_step_only", 0), e.get("use_graph"), e.get("batches_per_epoch"), e.get("batch")))
PY
done
