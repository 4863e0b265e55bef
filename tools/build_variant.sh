#!/bin/bash
# Build an experimental variant of the library outside the tree: tools/build_variant.sh /tmp/lib.so [-Dflags...]
out=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-math-errno"
/opt/rocm/bin/hipcc $B -mllvm -amdgpu-mfma-vgpr-form "$@" -c $R/qfa_amd/csrc/qfa_capi.hip -o $out.capi.o || exit 1
if [ ! -f /tmp/qfa_k32_variant.o ] || [ $R/qfa_amd/csrc/qfa_step_kernels.h -nt /tmp/qfa_k32_variant.o ]; then
  /opt/rocm/bin/hipcc $B -c $R/qfa_amd/csrc/qfa_k32.hip -o /tmp/qfa_k32_variant.o || exit 1
fi
# the XDL pass 2 / writer translation unit is taken from the tree's last build (tools/build_gx_variant.sh varies that one)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out.capi.o /tmp/qfa_k32_variant.o $R/qfa_amd/csrc/build/qfa_gx.o -o $out
