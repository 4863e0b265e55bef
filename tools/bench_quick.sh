#!/bin/bash
# GPU box: parity tests + both bench configs, short summary on stdout, full logs in gpurun_out/
tag=${1:-x}
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/t_$tag.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/t_$tag.log | cut -c1-200
for c in c2 c3; do
  timeout -k 10 400 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b_${c}_$tag.json 2> gpurun_out/b_${c}_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b_${c}_$tag.json"))
    print("$c", "%.3g spectra/s"%d["value"], "%.3f ms/step"%d["ms_per_step"], {k: round(v,3) for k,v in d["stage_ms"].items()}, "frac %.3f"%d["roofline"]["frac"], "step fp32 frac %.3f"%d["step_roofline"]["achieved_fp32_frac"])
except Exception as e:
    print("$c failed", e); print(open("gpurun_out/b_${c}_$tag.err").read()[-1500:])
PY
done
