#!/bin/bash
# GPU box: the round's evidence in one pass -- full gpu test suite, accuracy report, bench lines of every config, rocprofv3
# kernel-trace stats + PMC traffic of the headline config.  Everything lands under gpurun_out/ (copied into profiles/ afterwards).
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=400 2>&1 | tee gpurun_out/r3_pytest_gpu.log | tail -n 4
echo "== accuracy"; timeout -k 10 600 python tools/accuracy_report.py --full 2> gpurun_out/r3_accuracy.err | grep -v amdgpu.ids > gpurun_out/r3_accuracy.txt; tail -n 3 gpurun_out/r3_accuracy.txt
(python tools/scalar_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_scalar_probe_final.txt) || true
for cfg in c3 c2 c5 desi c1b; do
  echo "== bench $cfg"; timeout -k 10 400 python bench.py --config $cfg > gpurun_out/r3_bench_$cfg.json 2> gpurun_out/r3_bench_$cfg.err || { echo "bench $cfg failed"; tail -3 gpurun_out/r3_bench_$cfg.err; }
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3_bench_$cfg.json")); f=d.get("factored_z",{})
    print("$cfg %.4g spectra/s %.3f ms/step"%(d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["stage_ms"].items()}, "roofline.frac %.3f"%d["roofline"]["frac"], "| factored z %.3f ms"%f.get("ms_per_step",0), "| predict", round(d.get("predict",{}).get("ms_per_call",0),3))
except Exception as e: print("$cfg", e)
PY
done
echo "== bench c3 --flags 4 (three-product stage 3) and deterministic"
timeout -k 10 300 python bench.py --flags 4 --no-cpu-baseline --no-predict > gpurun_out/r3_bench_c3_fast.json 2>/dev/null
timeout -k 10 300 python bench.py --deterministic --no-cpu-baseline --no-predict > gpurun_out/r3_bench_c3_deterministic.json 2>/dev/null
echo "== profile"; tools/profile_round.sh r3 c3 > gpurun_out/r3_profile.log 2>&1; tail -n 25 gpurun_out/r3_profile.log
