#!/bin/bash
# GPU box: rocprofv3 PMC passes on bench.py (own runs, counters only + kernel-trace), aggregated per kernel.
# usage: tools/pmc.sh <tag> <config> "<counter list pass 1>" "<pass 2>" ...
tag=$1; cfg=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  echo "pmc pass: $set"
  i=$((i+1))
  d=$R/gpurun_out/pmc_${tag}_$i
  rm -rf $d
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-predict --sustain 0 ${BENCH_ARGS} > $d.json 2> $d.err || { echo "pass $i failed"; tail -5 $d.err; }
done
cd $R
python3 tools/pmc_agg.py gpurun_out/pmc_${tag}_ > gpurun_out/pmc_${tag}_summary.txt
rm -rf gpurun_out/pmc_${tag}_[0-9]*          # (raw counter dumps: the summary is what profiles/ keeps)
cat gpurun_out/pmc_${tag}_summary.txt
