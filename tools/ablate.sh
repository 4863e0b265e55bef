#!/bin/bash
# GPU box: timing-only ablation builds of the hot kernels (results are wrong by construction).
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  tools/build_variant.sh /tmp/libqfa_abl$a.so -DQFA_ABL=$a || exit 1
  QFA_HIP_LIB=/tmp/libqfa_abl$a.so timeout -k 10 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/abl_$a.json 2> gpurun_out/abl_$a.err
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$a.json")); print("ABL=$a", {k: round(v,3) for k,v in d["stage_ms"].items()})
PY
done
