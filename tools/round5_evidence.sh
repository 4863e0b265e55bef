#!/bin/bash
# GPU box: the round's evidence -- for EVERY bench config the bench line, rocprofv3 kernel-trace stats and the PMC
# passes that give HBM traffic per kernel (profiles/traffic_<cfg>.json carries the hash of the library it measured).  Run in two
# calls if the budget of one is short: tools/round5_evidence.sh "c3 c2"  and  tools/round5_evidence.sh "c5 desi c1b".
set -o pipefail
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
T=r5
declare -A NB=( [c3]=100000 [c2]=10000 [c5]=20000 [desi]=40000 [c1b]=500 [c4]=125000 )
for cfg in ${1:-c3 c2 c5 desi c1b}; do
  echo "== profile $cfg"; BENCH_ARGS="--no-epoch" tools/profile_round.sh ${T} $cfg > gpurun_out/${T}_profile_$cfg.log 2>&1; tail -n 4 gpurun_out/${T}_profile_$cfg.log
  python3 tools/make_traffic_json.py gpurun_out/pmc_${T}${cfg}_summary.txt $cfg ${NB[$cfg]} gpurun_out/traffic_$cfg.json > /dev/null && cp gpurun_out/traffic_$cfg.json profiles/traffic_$cfg.json && echo "traffic_$cfg.json written"
  echo "== bench $cfg"; timeout -k 10 400 python bench.py --config $cfg > gpurun_out/${T}_bench_$cfg.json 2> gpurun_out/${T}_bench_$cfg.err || { echo "bench $cfg failed"; tail -3 gpurun_out/${T}_bench_$cfg.err; }
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_bench_$cfg.json")); f=d.get("factored_z",{}); e=d.get("epoch",{}); z=d.get("zabs_kernels",{})
    print("$cfg %.4g spectra/s %.3f ms/step"%(d["value"], d["ms_per_step"]), {k: round(v,3) for k,v in d["stage_ms"].items() if isinstance(v,float)}, "roofline.frac %.3f traffic %s ratio %s"%(d["roofline"]["frac"], d["roofline"]["traffic"], d["step_roofline"]["traffic_ratio"]), "| zabs kernels %.3f ms"%z.get("ms_per_step",0), "| factored z %.3f ms"%f.get("ms_per_step",0), "| epoch %.3f ms/step"%e.get("ms_per_step",0), "| predict", round(d.get("predict",{}).get("ms_per_call",0),3))
except Exception as e: print("$cfg", e)
PY
done
