#!/bin/bash
# GPU box: diagnostic build with s_memtime stamps in k_grads; prints per-tile cycle shares.
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-math-errno -mllvm -amdgpu-mfma-vgpr-form -DQFA_ABL=7 $EXTRA qfa_amd/csrc/qfa_capi.hip -o /tmp/libqfa_abl7.so || exit 1
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
for w in range(4):
    n = buf[w * 8 + 6]
    print("wave", w, "tiles", n, {names[i]: round(buf[w * 8 + i] / max(n, 1)) for i in range(6)}, "sum", round(sum(buf[w*8+i] for i in range(6)) / max(n, 1)))
PY
