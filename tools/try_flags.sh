#!/bin/bash
# GPU box: build variants with extra -D flags and time bench c3.  usage: tools/try_flags.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
i=0
for fl in "$@"; do
  i=$((i+1))
  tools/build_variant.sh /tmp/libqfa_try$i.so $fl || exit 1
  QFA_HIP_LIB=/tmp/libqfa_try$i.so timeout -k 10 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/try_$i.json 2> gpurun_out/try_$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/try_$i.json")); print("[$fl]", "%.3f ms/step"%d["ms_per_step"], {k: round(v,3) for k,v in d["stage_ms"].items()})
PY
done
