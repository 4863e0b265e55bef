#!/bin/bash
# GPU box: s_memtime shares of the tile steps of wave 0 of one workgroup of k_moments_x (library variant built with
# tools/build_capi_variant.sh p1st -DQFA_P1_STAMPS=1).  usage: tools/p1_stamps.sh <variant name>
cd $GRAFT_REPO_ROOT
QFA_STAMP_LIB=$PWD/qfa_amd/libqfa_$1.so python - <<'PY'
import ctypes, sys, os, runpy
sys.path.insert(0, os.getcwd())
from qfa_amd import _lib
_lib.LIB_PATH = os.environ["QFA_STAMP_LIB"]
sys.argv = ["bench.py", "--config", "c3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-predict", "--sustain", "0"] + os.environ.get("STAMP_BENCH_ARGS", "").split()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
h = ctypes.CDLL(os.environ["QFA_STAMP_LIB"])
buf = (ctypes.c_ulonglong * 16)()
print("rc", h.qfa_p1_debug_stamps(buf))
for name, v in (("red tiles", list(buf[:8])), ("blue tiles", list(buf[8:]))):
    n = 1
    print(f"{name}: total cycles of the wave: land {v[1]}  weights {v[2]}  image dma issue {v[7]}  spectra loads issue {v[3]}  mfmas {v[4]}  dma wait {v[5]}  barrier {v[6]}  sum {sum(v[:8])}")
PY
