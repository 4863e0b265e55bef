#!/bin/bash
# build a variant of the library that differs in qfa_capi.o only: tools/build_capi_variant.sh <name> <extra hipcc flags>
set -e
cd "$(dirname "$0")/../qfa_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-math-errno -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize "$@" -c qfa_capi.hip -o /tmp/qfa_capi_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/qfa_capi_$name.o build/qfa_k32.o build/qfa_gx.o build/qfa_gt.o -o ../libqfa_$name.so
echo built ../libqfa_$name.so
