// Where do float atomic adds execute, and at what rate?  (pass 2 flushes 640 sums per tile-step and CU.)
// Every workgroup adds 1.0f K times per thread to a 336-KB array: the SHARED one with device-scope atomics (what
// atomicAdd() does: executed at the memory side), or the array of ITS XCD (HW_REG_XCC_ID) with workgroup-scope atomics
// (no sc1 bit: executed in that XCD's L2, which every adder of that array shares).  Prints the time per launch and
// whether the total over all arrays equals the number of adds (a lost update would show).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int M = 84000, K = 64;
template <int MODE>
__global__ void k(float *buf, int *xcc_seen) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;      // HW_REG_XCC_ID, bits 3:0
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1 << xcc);
    float *b = MODE == 0 ? buf : buf + (size_t)xcc * M;
    unsigned o = (blockIdx.x * 2654435761u) % M;
    for (int i = 0; i < K; ++i) {
        const unsigned idx = (o + i * 256 + threadIdx.x) % M;
        if (MODE == 0) atomicAdd(b + idx, 1.0f);
        else __hip_atomic_fetch_add(b + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
int main() {
    float *d; int *seen;
    hipMalloc(&d, 8 * M * sizeof(float)); hipMalloc(&seen, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 4096;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0, 8 * M * sizeof(float)); hipMemset(seen, 0, 4);
            hipEventRecord(e0);
            if (mode == 0) k<0><<<grid, 256>>>(d, seen); else k<1><<<grid, 256>>>(d, seen);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<float> h(8 * M); int s;
            hipMemcpy(h.data(), d, 8 * M * sizeof(float), hipMemcpyDeviceToHost); hipMemcpy(&s, seen, 4, hipMemcpyDeviceToHost);
            double tot = 0; for (float v : h) tot += v;
            printf("%s: %.3f ms for %.1f M adds (%.1f G adds/s), sum %.0f expected %.0f, xcc mask %x\n",
                   mode ? "workgroup scope, per-XCD arrays" : "device scope, one array        ", ms,
                   grid * 256.0 * K / 1e6, grid * 256.0 * K / ms / 1e6, tot, grid * 256.0 * K, s);
        }
    }
    return 0;
}
