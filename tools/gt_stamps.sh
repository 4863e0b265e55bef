#!/bin/bash
# GPU box: s_memtime shares of the walk of waves 0 and 4 of one workgroup of k_grads_t (library variant built with
# tools/build_gt_variant.sh st -DQFA_GT_STAMPS=1).  usage: tools/gt_stamps.sh <variant name>
cd $GRAFT_REPO_ROOT
QFA_STAMP_LIB=$PWD/qfa_amd/libqfa_$1.so python - <<'PY'
import ctypes, sys, os, runpy
sys.path.insert(0, os.getcwd())
from qfa_amd import _lib
_lib.LIB_PATH = os.environ["QFA_STAMP_LIB"]
sys.argv = ["bench.py", "--config", "c3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-predict", "--sustain", "0", "--flags", "64"] + os.environ.get("STAMP_BENCH_ARGS", "").split()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
h = ctypes.CDLL(os.environ["QFA_STAMP_LIB"])
buf = (ctypes.c_ulonglong * 32)()
print("rc", h.qfa_gt_debug_stamps(buf))
for name, v in (("wave 0", list(buf[:16])), ("wave 4", list(buf[16:]))):
    n = max(v[7], 1)
    print(f"{name}: groups {v[7]}; cycles per group: dma wait {v[0]/n:.0f}  barrier {v[1]/n:.0f}  dma issue {v[2]/n:.0f}  stage 1 {v[5]/n:.0f}  take {v[6]/n:.0f}  stage 2 {v[3]/n:.0f}  stage 3 {v[4]/n:.0f}  sum {sum(v[:7])/n:.0f}")
PY
