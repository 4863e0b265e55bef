#!/bin/bash
# GPU box: predict-call times of library variants back to back on one box.  usage: tools/ab_predict.sh <variant> ... ("default" = shipped)
for v in "$@"; do
  if [ "$v" != "default" ]; then L="qfa_amd/libqfa_$v.so"; else L="qfa_amd/libqfa_hip.so"; fi
  timeout -k 10 200 python tools/with_lib.py $L bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-epoch --sustain 0 ${BENCH_ARGS} > gpurun_out/abp_$v.json 2> gpurun_out/abp_$v.err || { echo "$v failed"; tail -3 gpurun_out/abp_$v.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abp_$v.json")); p=d["predict"]; q=d.get("predict_sdss_shape")
print("%-10s predict %.3f ms/call" % ("$v", p["ms_per_call"]), {k: round(x,3) for k,x in p["stage_ms"].items()}, "writer %.0f GB/s" % p["roofline"]["achieved"], ("| sdss shape writer %.3f ms %.0f GB/s" % (q["stage_ms"]["writer"], q["roofline"]["achieved"])) if q else "")
PY
done
