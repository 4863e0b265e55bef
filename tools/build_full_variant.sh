#!/bin/bash
# build a variant of the library whose qfa_capi.o AND qfa_gx.o carry extra flags: tools/build_full_variant.sh <name> <extra hipcc flags>
set -e
cd "$(dirname "$0")/../qfa_amd/csrc"
name=$1; shift
B="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-math-errno -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc $B -fno-slp-vectorize "$@" -c qfa_capi.hip -o /tmp/qfa_capi_$name.o &
/opt/rocm/bin/hipcc $B -mllvm -amdgpu-sched-strategy=iterative-maxocc "$@" -c qfa_gx.hip -o /tmp/qfa_gx_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/qfa_capi_$name.o build/qfa_k32.o build/qfa_gt.o /tmp/qfa_gx_$name.o -o ../libqfa_$name.so
echo built ../libqfa_$name.so
