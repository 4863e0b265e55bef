// What does a DEPENDENT accumulation chain of v_mfma_f32_16x16x32_bf16 cost against independent ones?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_chain.hip -o tools/ubench/mfma_chain && tools/ubench/mfma_chain
// One wave issues 96 MFMAs as C chains (C = 1, 2, 3, 4, 6: MFMA i accumulates into chain i % C); cycles from s_memtime
// (100 MHz reference clock scaled by the measured ratio is avoided: the table prints s_memtime ticks of the whole loop and
// the ratio to the 1-chain case).  Then the same with 2 and 4 waves on one SIMD (256- and 512-thread blocks... a block of
// 64 W threads puts W / 4 waves on each SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ void k(const float *in, float *out, unsigned long long *ticks, int reps) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[lane * 8 + j]; b[j] = (__bf16)in[512 + lane * 8 + j]; }
    f32x4 c[C];
    for (int i = 0; i < C; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 96; ++i) c[i % C] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i % C], 0, 0, 0);
    }
    for (int i = 0; i < C; ++i) asm volatile("" ::"v"(c[i]));
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < C; ++i) s += c[i][0];
    out[threadIdx.x] = s;
    if (lane == 0) { ticks[2 * (threadIdx.x >> 6)] = t0; ticks[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int C>
static double run(int threads, const float *in, float *out, unsigned long long *tk) {
    const int reps = 200;
    hipLaunchKernelGGL(k<C>, 1, threads, 0, 0, in, out, tk, reps);
    hipLaunchKernelGGL(k<C>, 1, threads, 0, 0, in, out, tk, reps);
    unsigned long long t[32];
    hipMemcpy(t, tk, 16 * (threads / 64), hipMemcpyDeviceToHost);
    unsigned long long lo = t[0], hi = t[1];                 // the block's span: the arbiter serves the oldest wave first
    for (int w = 1; w < threads / 64; ++w) { lo = t[2 * w] < lo ? t[2 * w] : lo; hi = t[2 * w + 1] > hi ? t[2 * w + 1] : hi; }
    return (double)(hi - lo) / (reps * 96.0);
}

int main() {
    float *in, *out;
    unsigned long long *tk;
    hipMalloc(&in, 4096); hipMalloc(&out, 4096); hipMalloc(&tk, 512);
    hipMemset(in, 0, 4096);
    printf("s_memtime ticks of the block (first start to last end) per MFMA of ONE wave\n");
    for (int threads : {64, 256, 512, 1024}) {
        const double c1 = run<1>(threads, in, out, tk), c2 = run<2>(threads, in, out, tk), c3 = run<3>(threads, in, out, tk),
                     c4 = run<4>(threads, in, out, tk), c6 = run<6>(threads, in, out, tk);
        printf("%4d threads (%d wave(s) per SIMD): 1 chain %.4f  2 chains %.4f  3 chains %.4f  4 chains %.4f  6 chains %.4f   (1 : 2 : 4 = %.2f : %.2f : 1)\n",
               threads, threads / 256 > 0 ? threads / 256 : 1, c1, c2, c3, c4, c6, c1 / c4, c2 / c4);
    }
    return 0;
}
