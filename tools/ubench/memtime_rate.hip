// What does s_memtime count?  Ticks of one wave against the wall clock (HIP events), for one workgroup on an idle card and
// for a card filled with MFMA work; and MFMAs per tick, which is 1/16 if a tick is a shader cycle.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/memtime_rate.hip -o tools/ubench/memtime_rate && tools/ubench/memtime_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *ticks, int reps) {
    u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r)
        asm volatile(".rept 64\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n .endr" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
    asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[0];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; unsigned long long *tk;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&tk, 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256, 1024, 4096}) {
        for (int reps : {2000, 20000, 100000}) {
            hipLaunchKernelGGL(k, blocks, 256, 0, 0, out, tk, 10);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, blocks, 256, 0, 0, out, tk, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
            const double mf = 128.0 * reps;
            printf("%5d blocks x 4 waves, %6d reps: kernel %.3f ms, block 0: %llu ticks (%.3f ticks/ns if it ran the whole time), %.2f ticks per MFMA, %.2f ns per MFMA\n",
                   blocks, reps, ms, t, (double)t / (ms * 1e6), (double)t / mf, ms * 1e6 / mf / (blocks > 256 ? (blocks / 256.0) : 1.0));
        }
    }
    return 0;
}
