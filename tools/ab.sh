#!/bin/bash
# GPU box: A/B two library builds on the same bench configs.  usage: tools/ab.sh libA.so libB.so "<bench args>" ...
A=$1; B=$2; shift 2
for args in "$@"; do
  for lib in $A $B; do
    QFA_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py $args --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab.json")); print("$lib | $args |", "%.3f ms/step"%d["ms_per_step"], {k: round(v,3) for k,v in d["stage_ms"].items()})
except Exception as e:
    print("$lib failed", e); print(open("gpurun_out/ab.err").read()[-800:])
PY
  done
done
