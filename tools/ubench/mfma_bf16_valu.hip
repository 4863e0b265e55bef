// Does VALU work of the SAME wave hide under v_mfma_f32_16x16x32_bf16 (16 cycles on the XDL pipe)?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_bf16_valu.hip -o tools/ubench/mfma_bf16_valu && tools/ubench/mfma_bf16_valu
// Per slot: one MFMA (two accumulator chains alternate) followed by n independent v_fma_f32 (n = 0..8) or by n/2 v_exp_f32;
// s_memtime ticks of the block (first start to last end) per slot of one wave, for 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FMA1 "v_fma_f32 %4, %4, %8, %9\n"
#define FMA2 FMA1 "v_fma_f32 %5, %5, %8, %9\n"
#define FMA3 FMA2 "v_fma_f32 %6, %6, %8, %9\n"
#define FMA4 FMA3 "v_fma_f32 %7, %7, %8, %9\n"
#define FMA6 FMA4 "v_fma_f32 %4, %4, %9, %8\n v_fma_f32 %5, %5, %9, %8\n"
#define FMA8 FMA6 "v_fma_f32 %6, %6, %9, %8\n v_fma_f32 %7, %7, %9, %8\n"
#define EXP2 "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
#define SLOT(V) "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n" V "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n" V
#define RUN(V) asm volatile(".rept 48\n" SLOT(V) ".endr" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(p), "v"(q))

template <int MODE>
__global__ void k(const float *in, float *out, unsigned long long *ticks, int reps) {
    const int lane = threadIdx.x & 63;
    u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float x0 = in[lane], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, p = 0.5f, q = 0.25f;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) RUN("");
        else if (MODE == 1) RUN(FMA1);
        else if (MODE == 2) RUN(FMA2);
        else if (MODE == 3) RUN(FMA3);
        else if (MODE == 4) RUN(FMA4);
        else if (MODE == 6) RUN(FMA6);
        else if (MODE == 8) RUN(FMA8);
        else if (MODE == 9) RUN(EXP2);
        else if (MODE == 10) asm volatile(".rept 96\n" FMA4 ".endr" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(p), "v"(q));
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[threadIdx.x] = c0[0] + c1[0] + x0 + x1 + x2 + x3;
    if (lane == 0) { ticks[2 * (threadIdx.x >> 6)] = t0; ticks[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int MODE>
static double run(int threads, const float *in, float *out, unsigned long long *tk) {
    const int reps = 100;
    hipLaunchKernelGGL(k<MODE>, 1, threads, 0, 0, in, out, tk, reps);
    hipLaunchKernelGGL(k<MODE>, 1, threads, 0, 0, in, out, tk, reps);
    unsigned long long t[32];
    hipMemcpy(t, tk, 16 * (threads / 64), hipMemcpyDeviceToHost);
    unsigned long long lo = t[0], hi = t[1];
    for (int w = 1; w < threads / 64; ++w) { lo = t[2 * w] < lo ? t[2 * w] : lo; hi = t[2 * w + 1] > hi ? t[2 * w + 1] : hi; }
    return (double)(hi - lo) / (reps * 96.0);
}
int main() {
    float *in, *out;
    unsigned long long *tk;
    hipMalloc(&in, 4096); hipMalloc(&out, 4096); hipMalloc(&tk, 512);
    hipMemset(in, 0, 4096);
    for (int threads : {256, 512}) {
        printf("%d wave(s) per SIMD, ticks per slot: MFMA alone %.1f | +1 fma %.1f | +2 %.1f | +3 %.1f | +4 %.1f | +6 %.1f | +8 %.1f | +2 v_exp %.1f | 4 fma alone %.1f\n",
               threads / 256, run<0>(threads, in, out, tk), run<1>(threads, in, out, tk), run<2>(threads, in, out, tk), run<3>(threads, in, out, tk),
               run<4>(threads, in, out, tk), run<6>(threads, in, out, tk), run<8>(threads, in, out, tk), run<9>(threads, in, out, tk), run<10>(threads, in, out, tk));
    }
    return 0;
}
