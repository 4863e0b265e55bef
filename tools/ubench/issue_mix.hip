// What does a NON-vector instruction cost a wave that mixes MFMAs and VALU work?  (round 5: k_grads_t executes per group of 16
// spectra ~200 scalar + ~70 wait / nop instructions beside 259 VALU + 87 MFMA + 72 LDS; DESIGN.md priced only the vector ones.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/issue_mix.hip -o tools/ubench/issue_mix && tools/ubench/issue_mix
// Per slot: one v_mfma_f32_16x16x32_bf16 (two accumulator chains alternate) + 2 independent v_fma_f32 + n instructions of the class
// under test (n = 0, 2, 4, 8): s_add_u32 on four SGPRs, s_nop 0, a satisfied s_waitcnt, s_mov_b32 m0, ds_read_b128 (conflict-free).
// s_memtime ticks per slot of one wave, for 1 and 2 waves per SIMD (512 threads = two waves on every SIMD of the CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FMA2 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
#define SA1 "s_add_u32 s20, s20, 1\n"
#define SA2 SA1 "s_add_u32 s21, s21, 1\n"
#define SA4 SA2 "s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
#define SA8 SA4 SA4
#define NP2 "s_nop 0\n s_nop 0\n"
#define NP4 NP2 NP2
#define NP8 NP4 NP4
#define WT2 "s_waitcnt vmcnt(0)\n s_waitcnt lgkmcnt(15)\n"
#define WT4 WT2 WT2
#define WT8 WT4 WT4
#define M02 "s_mov_b32 m0, s20\n s_mov_b32 m0, s21\n"
#define M04 M02 M02
#define M08 M04 M04
#define DS2 "ds_read_b128 %10, %12\n ds_read_b128 %11, %12 offset:1024\n"
#define DS4 DS2 DS2
#define DS8 DS4 DS4
#define SLOT(V) "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n" FMA2 V "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n" FMA2 V
#define RUN(V)                                                                                                                      \
    asm volatile(".rept 48\n" SLOT(V) ".endr\n s_waitcnt lgkmcnt(0)"                                                                \
                 : "+v"(c0), "+v"(c1)                                                                                               \
                 : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(p), "v"(q), "v"(d0), "v"(d1), "v"(la)                      \
                 : "s20", "s21", "s22", "s23", "memory")

template <int MODE>
__global__ void k(const float *in, float *out, unsigned long long *ticks, int reps) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 2048];
    const int lane = threadIdx.x & 63;
    u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float x0 = in[lane], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, p = 0.5f, q = 0.25f;
    const unsigned la = (unsigned)(size_t)(lds + (threadIdx.x >> 6) * 2048 + lane * 16);
    reinterpret_cast<u32x4 *>(lds)[threadIdx.x] = u32x4{0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) RUN("");
        else if (MODE == 1) RUN(SA2);
        else if (MODE == 2) RUN(SA4);
        else if (MODE == 3) RUN(SA8);
        else if (MODE == 4) RUN(NP4);
        else if (MODE == 5) RUN(NP8);
        else if (MODE == 6) RUN(WT4);
        else if (MODE == 7) RUN(WT8);
        else if (MODE == 8) RUN(M04);
        else if (MODE == 9) RUN(M08);
        else if (MODE == 10) RUN(DS2);
        else if (MODE == 11) RUN(DS4);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[threadIdx.x] = c0[0] + c1[0] + x0 + x1 + x2 + x3 + __uint_as_float(d0[0] + d1[0]);
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
        printf("%d wave(s) per SIMD, s_memtime ticks per slot (MFMA + 2 v_fma + n x):\n", threads / 256);
        printf("  nothing %.1f | s_add x2 %.1f x4 %.1f x8 %.1f | s_nop x4 %.1f x8 %.1f | s_waitcnt x4 %.1f x8 %.1f | s_mov m0 x4 %.1f x8 %.1f | ds_read_b128 x2 %.1f x4 %.1f\n",
               run<0>(threads, in, out, tk), run<1>(threads, in, out, tk), run<2>(threads, in, out, tk), run<3>(threads, in, out, tk),
               run<4>(threads, in, out, tk), run<5>(threads, in, out, tk), run<6>(threads, in, out, tk), run<7>(threads, in, out, tk),
               run<8>(threads, in, out, tk), run<9>(threads, in, out, tk), run<10>(threads, in, out, tk), run<11>(threads, in, out, tk));
    }
    return 0;
}
