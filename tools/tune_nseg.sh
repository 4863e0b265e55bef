#!/bin/bash
# GPU box: pixel-segment counts of pass 1 / pass 2 by hand (build with -DQFA_TUNE_ENV)
cd $GRAFT_REPO_ROOT
tools/build_variant.sh /tmp/libqfa_tune.so -DQFA_TUNE_ENV || exit 1
for v in "$@"; do
  n1=${v%%:*}; n2=${v##*:}
  QFA_NSEG1=$n1 QFA_NSEG2=$n2 QFA_HIP_LIB=/tmp/libqfa_tune.so timeout -k 10 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/tune.json 2> gpurun_out/tune.err
  python - <<PY
import json
d=json.load(open("gpurun_out/tune.json")); print("nseg1=$n1 nseg2=$n2", "%.3f ms/step"%d["ms_per_step"], {k: round(v,3) for k,v in d["stage_ms"].items()})
PY
done
