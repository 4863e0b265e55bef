#!/bin/bash
# build a variant of the library that differs in qfa_gt.o only: tools/build_gt_variant.sh <name> <extra hipcc flags>
set -e
cd "$(dirname "$0")/../qfa_amd/csrc"
name=$1; shift
mkdir -p build/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-math-errno -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=iterative-maxocc -fno-slp-vectorize "$@" -save-temps=obj -c qfa_gt.hip -o build/var_$name/qfa_gt.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/qfa_capi.o build/qfa_k32.o build/qfa_gx.o build/var_$name/qfa_gt.o -o ../libqfa_$name.so
grep -E "scratch_" build/var_$name/qfa_gt-hip-amdgcn-amd-amdhsa-gfx950.s | head -3
echo built ../libqfa_$name.so
