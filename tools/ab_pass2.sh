#!/bin/bash
# GPU box: pass-2 time of library variants (tools/build_gt_variant.sh etc.), zabs form and factored-z form, back to back on one box.
# usage: tools/ab_pass2.sh <variant> ...   ("default" = the shipped library; else qfa_amd/libqfa_<variant>.so, same ABI)
for v in "$@"; do
  if [ "$v" != "default" ]; then L="qfa_amd/libqfa_$v.so"; else L="qfa_amd/libqfa_hip.so"; fi
  timeout -k 10 120 python tools/with_lib.py $L bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-predict --no-epoch --sustain 0 ${BENCH_ARGS} > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v failed"; tail -3 gpurun_out/ab_$v.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
f=d.get("factored_z") or {"stage_ms":{"pass1_moments":0,"pass2_grads":0},"ms_per_step":0}
print("%-10s zabs: step %.3f p1 %.3f p2 %.3f | zfac: step %.3f p1 %.3f p2 %.3f" % ("$v", d["ms_per_step"], d["stage_ms"]["pass1_moments"], d["stage_ms"]["pass2_grads"], f["ms_per_step"], f["stage_ms"]["pass1_moments"], f["stage_ms"]["pass2_grads"]))
PY
done
