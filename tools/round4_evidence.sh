#!/bin/bash
# GPU box: the round's evidence in one pass -- full gpu test suite, accuracy report, bench lines of every config (step legs,
# epoch leg, predict, CPU baseline), rocprofv3 kernel-trace stats + PMC traffic of the headline config.  Everything lands under
# gpurun_out/ (copied into profiles/ afterwards).
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
T=r4
echo "== pytest -m gpu"; timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=400 2>&1 | tee gpurun_out/${T}_pytest_gpu.log | tail -n 4
echo "== accuracy"; timeout -k 10 600 python tools/accuracy_report.py --full 2> gpurun_out/${T}_accuracy.err | grep -v amdgpu.ids > gpurun_out/${T}_accuracy.txt; tail -n 3 gpurun_out/${T}_accuracy.txt
for cfg in c3 c2 c5 desi c1b; do
  echo "== bench $cfg"; timeout -k 10 400 python bench.py --config $cfg > gpurun_out/${T}_bench_$cfg.json 2> gpurun_out/${T}_bench_$cfg.err || { echo "bench $cfg failed"; tail -3 gpurun_out/${T}_bench_$cfg.err; }
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench_$cfg.json")); f=d.get("factored_z",{}); e=d.get("epoch",{})
    print("$cfg %.4g spectra/s %.3f ms/step"%(d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["stage_ms"].items()}, "roofline.frac %.3f"%d["roofline"]["frac"], "| factored z %.3f ms"%f.get("ms_per_step",0), "| epoch %.3f ms/step (%.3f of step)"%(e.get("ms_per_step",0), e.get("vs_step_only",0)), "| predict", round(d.get("predict",{}).get("ms_per_call",0),3))
except Exception as e: print("$cfg", e)
PY
done
echo "== bench c4 at N = 1 through RCCL (QFA_BENCH_FORCE_DIST=1)"
QFA_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --config c4 --no-cpu-baseline --no-predict --no-epoch > gpurun_out/${T}_bench_c4_n1_rccl.json 2>/dev/null
echo "== bench c3 deterministic"
timeout -k 10 300 python bench.py --deterministic --no-cpu-baseline --no-predict --no-epoch > gpurun_out/${T}_bench_c3_deterministic.json 2>/dev/null
echo "== profile"; BENCH_ARGS="--no-epoch" tools/profile_round.sh ${T} c3 > gpurun_out/${T}_profile.log 2>&1; tail -n 25 gpurun_out/${T}_profile.log
python3 tools/make_traffic_json.py gpurun_out/pmc_${T}c3_summary.txt c3 100000 gpurun_out/traffic_c3.json > /dev/null && echo "traffic_c3.json written"
