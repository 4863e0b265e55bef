#!/bin/bash
# GPU box: bench.py (c3, pass timings) for each library variant qfa_amd/libqfa_<name>.so given on the command line
for v in "$@"; do
  QFA_HIP_LIB=$PWD/qfa_amd/libqfa_$v.so timeout -k 10 200 python bench.py --steps 10 --warmup 5 --sustain 0 --no-predict --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in d['stage_ms'].items()})"
done
