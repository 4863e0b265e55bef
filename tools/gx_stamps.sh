#!/bin/bash
# GPU box: s_memtime shares of one workgroup's tile steps in k_grads_x (library variant built with -DQFA_GX_STAMPS=1:
# tools/build_gx_variant.sh st -DQFA_GX_STAMPS=1).  usage: tools/gx_stamps.sh <variant name>
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
buf = (ctypes.c_ulonglong * 64)()
print("rc", h.qfa_gx_debug_stamps(buf))
a, b = list(buf[:32]), list(buf[32:])
n = a[30]
print("role A, tiles", n, " idle steps", a[31])
for tag, off, cnt in (("blue", 0, None), ("red", 16, None)):
    for hh in range(2):
        v = a[off + 8 * hh: off + 8 * hh + 6]
        print(f"  {tag} h{hh}: vmcnt wait {v[5]}  take {v[0]}  stage1 {v[1]}  stage2 {v[2]}  reduce+store {v[3]}  barrier {v[4]}   (total cycles over all tiles)")
print("  sum", sum(a[:30]), "per tile", sum(a[:30]) / max(n, 1))
print("role B")
for hh in range(2):
    v = b[8 * hh: 8 * hh + 4]
    print(f"  h{hh}: dma issue+flush {v[0]}  stage3 part {v[1]}  vmcnt wait {v[2]}  barrier {v[3]}")
print("  sum", sum(b[:30]), "per tile", sum(b[:30]) / max(n, 1))
PY
