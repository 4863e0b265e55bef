#!/bin/bash
# GPU box: step and posterior-predict times of library variants back to back on one box.
# usage: [BENCH_ARGS="--config desi"] tools/ab_predict.sh <variant> ...   ("default" = the shipped library)
for v in "$@"; do
  if [ "$v" != "default" ]; then export QFA_HIP_LIB="$PWD/qfa_amd/libqfa_$v.so"; else unset QFA_HIP_LIB; fi
  timeout -k 10 180 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --sustain 0 ${BENCH_ARGS} > gpurun_out/abp_$v.json 2> gpurun_out/abp_$v.err || { echo "$v failed"; tail -3 gpurun_out/abp_$v.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abp_$v.json")); p=d["predict"]; s=p["stage_ms"]
print("%-10s step %.3f p1 %.3f p2 %.3f | predict %.3f: pass1 %.3f solve %.3f writer %.3f (%.0f GB/s)" % ("$v", d["ms_per_step"], d["stage_ms"]["pass1_moments"], d["stage_ms"]["pass2_grads"], p["ms_per_call"], s["images_and_pass1"], s["solve"], s["writer"], p["roofline"]["achieved"]))
PY
done
