// Two (or four) waves on one SIMD with DIFFERENT instruction streams: what does a wave's VALU stream cost beside another
// wave's v_mfma_f32_16x16x32_bf16 stream?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_roles.hip -o tools/ubench/mfma_roles && tools/ubench/mfma_roles
// A block of 256 W threads puts W waves on every SIMD (wave w -> SIMD w % 4); role of a wave = role[w / 4].
// roles: 'M' = 96 MFMAs per rep (two chains), 'F' = 384 independent v_fma_f32 per rep, 'X' = 96 x (MFMA + 4 fma) per rep,
// 'E' = 192 v_exp_f32 per rep, '-' = idle.  Prints every role's own ticks per rep / 96.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FMA4 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define OPS : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(p), "v"(q)
__global__ void k(const float *in, float *out, unsigned long long *ticks, int reps, int r0, int r1, int r2, int r3) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int roles[4] = {r0, r1, r2, r3};
    const int role = roles[w >> 2];
    u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float x0 = in[lane], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, p = 0.5f, q = 0.25f;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (role == 'M')
        for (int r = 0; r < reps; ++r) asm volatile(".rept 48\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n .endr" OPS);
    else if (role == 'F')
        for (int r = 0; r < reps; ++r) asm volatile(".rept 96\n" FMA4 ".endr" OPS);
    else if (role == 'X')
        for (int r = 0; r < reps; ++r) asm volatile(".rept 48\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n" FMA4 "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n" FMA4 ".endr" OPS);
    else if (role == 'E')
        for (int r = 0; r < reps; ++r) asm volatile(".rept 96\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n .endr" OPS);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[threadIdx.x] = c0[0] + c1[0] + x0 + x1 + x2 + x3;
    if (lane == 0) ticks[w] = t1 - t0;
}
static void run(const char *roles, const float *in, float *out, unsigned long long *tk) {
    const int W = (int)strlen(roles), reps = 100;
    int r[4] = {'-', '-', '-', '-'};
    for (int i = 0; i < W; ++i) r[i] = roles[i];
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, 1, 256 * W, 0, 0, in, out, tk, reps, r[0], r[1], r[2], r[3]);
    unsigned long long t[16];
    hipMemcpy(t, tk, 8 * 4 * W, hipMemcpyDeviceToHost);
    printf("%-5s:", roles);
    for (int i = 0; i < W; ++i) printf("  wave %d (%c) %.1f", i, roles[i], (double)t[4 * i] / (reps * 96.0));
    printf("   ticks per 96th of a rep (M: per MFMA, F: per 4 fma, X: per MFMA + 4 fma, E: per 2 exp)\n");
}
int main() {
    float *in, *out;
    unsigned long long *tk;
    hipMalloc(&in, 4096); hipMalloc(&out, 8192); hipMalloc(&tk, 512);
    hipMemset(in, 0, 4096);
    for (const char *r : {"M", "F", "E", "X", "MM", "FF", "MF", "ME", "XX", "FFFF", "MFF", "MMFF", "MFFF", "XF"}) run(r, in, out, tk);
    return 0;
}
