#!/bin/bash
# GPU box: evidence for profiles/: (1) kernel-trace stats of the default bench, (2) PMC passes for HBM traffic.
tag=${1:-r1}; cfg=${2:-c3}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/prof_${tag}_${cfg}; rm -rf $d
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $d.json 2> $d.err
f=$(find $d -name "*kernel_stats.csv" | head -1)
(head -1 $f; grep -E '"(void )?k_' $f) > $R/gpurun_out/${tag}_${cfg}_kernel_stats.csv
cd $R
python3 tools/launch_durations.py $d $d.json > gpurun_out/${tag}_${cfg}_kernel_launches.txt
rm -rf $d          # (the raw trace: tens of MB per config; gpurun copies back at most 64 MiB)
tools/pmc.sh ${tag}${cfg} $cfg "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_EA0_ATOMIC" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_HIT TCC_MISS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" > /dev/null
cp gpurun_out/pmc_${tag}${cfg}_summary.txt gpurun_out/${tag}_${cfg}_pmc_summary.txt
cat gpurun_out/${tag}_${cfg}_kernel_stats.csv | cut -c1-150
grep -A12 "^k_grads\|^k_moments\|^k_solve\|^k_prep_pst" gpurun_out/${tag}_${cfg}_pmc_summary.txt | grep -v "^--" | head -60
