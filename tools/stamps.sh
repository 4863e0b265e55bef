#!/bin/bash
# GPU box: diagnostic build with s_memtime stamps in k_grads; prints per-tile cycle shares.
cd $GRAFT_REPO_ROOT
tools/build_variant.sh /tmp/libqfa_abl7.so -DQFA_ABL=7 $EXTRA || exit 1
QFA_HIP_LIB=/tmp/libqfa_abl7.so python - <<'PY'
import ctypes, json, subprocess, sys, os
sys.argv = ["bench.py", "--config", "c3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
from qfa_amd import _lib
h = _lib.lib()
buf = (ctypes.c_ulonglong * 64)()
print("rc", h.qfa_debug_stamps(buf))
names = ["issue loads", "region1 (stage2 + stage1 next)", "stage3", "tile store", "barrier", "flush"]
for seg in range(4):
    for tag, off in (("blue", 0), ("red", 8)):
        n = buf[seg * 16 + off + 6]
        if n:
            v = [buf[seg * 16 + off + i] for i in range(6)]
            print("segment", seg, tag, "tiles", n, {names[i]: round(v[i] / n) for i in range(6)}, "sum", round(sum(v) / n))
PY
