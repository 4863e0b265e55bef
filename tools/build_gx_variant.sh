#!/bin/bash
# build a variant of the library that differs in qfa_gx.o only: tools/build_gx_variant.sh <name> <extra hipcc flags>
set -e
cd "$(dirname "$0")/../qfa_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-math-errno -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-sched-strategy=iterative-maxocc "$@" -save-temps=obj -c qfa_gx.hip -o /tmp/qfa_gx_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/qfa_capi.o build/qfa_k32.o build/qfa_gt.o /tmp/qfa_gx_$name.o -o ../libqfa_$name.so
echo built ../libqfa_$name.so
