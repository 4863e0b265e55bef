#!/bin/bash
# GPU box: step time with pass 2 forced to k_grads_x (flags 0x2) and to k_grads_t (0x40) over batch sizes -- where does the automatic
# dispatch (qfa_host.h, pass2_use_pixres) have to switch?   usage: tools/pass2_crossover.sh "<config> <B> <B> ..." ...
for spec in "$@"; do
  set -- $spec; cfg=$1; shift
  for B in "$@"; do
    line="$cfg B=$B"
    for fl in 0x2 0x40; do
      timeout -k 10 120 python bench.py --config $cfg --batch $B --flags $fl --no-cpu-baseline --no-predict --no-epoch --sustain 0 > gpurun_out/xo.json 2> gpurun_out/xo.err || { line="$line | flags=$fl failed"; continue; }
      line="$line | $(python -c "
import json; d=json.load(open('gpurun_out/xo.json')); print('%s step %.4f p2 %.4f solve %.4f' % (d['roofline']['kernel'], d['ms_per_step'], d['stage_ms']['pass2_grads'], d['stage_ms']['solve']))")"
    done
    echo "$line"
  done
done
