// gfx950: what v_permlane16_swap / v_permlane32_swap exchange (k_grads_x sums across the four 16-lane rows with them).
// Prints lanes 0, 8, .. 56 of both results for a = lane, b = 1000 + lane: permlane16_swap gives [a.r0 b.r0 a.r2 b.r2] and
// [a.r1 b.r1 a.r3 b.r3] (r = row of 16 lanes), permlane32_swap gives [a.r0 a.r1 b.r0 b.r1] and [a.r2 a.r3 b.r2 b.r3].
#include <hip/hip_runtime.h>
__global__ void k(unsigned *o) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned *d; hipMalloc(&d, 1024); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; ++j) { for (int i = 0; i < 64; i += 8) printf("%u ", h[j * 64 + i]); printf("\n"); }
}
