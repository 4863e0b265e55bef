#!/bin/bash
# build a variant of the library that differs in qfa_k32.o only: tools/build_k32_variant.sh <name> <extra hipcc flags>
set -e
cd "$(dirname "$0")/../qfa_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-math-errno -Wall -Wno-unused-function "$@" -c qfa_k32.hip -o /tmp/qfa_k32_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/qfa_capi.o /tmp/qfa_k32_$name.o build/qfa_gx.o -o ../libqfa_$name.so
echo built ../libqfa_$name.so
