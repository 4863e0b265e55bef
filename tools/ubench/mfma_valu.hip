// Microbenchmark: how much VALU / transcendental / LDS work hides under v_mfma_f32_16x16x4_f32 on one
// gfx950 SIMD, from the same wave and from a second wave.  Prints cycles per MFMA slot.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP 64
#define OUTER 64

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, float seed) {
    __shared__ float lds[8192];
    lds[threadIdx.x] = seed;
    __syncthreads();
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
    float av = seed, bv = seed * 2;
    float l0 = 0, l1 = 0;
    f32x4 A4 = {seed, seed, seed, seed}, B4 = {seed, seed, seed, seed}, L0 = A4, L1 = A4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 A2 = {seed, seed}, B2 = {seed, seed};
    const float *lp = lds + (threadIdx.x & 63);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int o = 0; o < OUTER; ++o) {
        if (MODE == 0)          // MFMA only, two accumulator chains
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv));
        else if (MODE == 1)     // MFMA + 2 VALU
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3\n"
                         "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_fma_f32 %6, %6, %2, %3\n v_fma_f32 %7, %7, %2, %3\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        else if (MODE == 2)     // MFMA + 4 VALU
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3\n v_fma_f32 %6, %6, %2, %3\n v_fma_f32 %7, %7, %2, %3\n"
                         "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n v_fma_f32 %10, %10, %2, %3\n v_fma_f32 %11, %11, %2, %3\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        else if (MODE == 3)     // MFMA + 6 VALU
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3\n v_fma_f32 %6, %6, %2, %3\n v_fma_f32 %7, %7, %2, %3\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n"
                         "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n v_fma_f32 %10, %10, %2, %3\n v_fma_f32 %11, %11, %2, %3\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        else if (MODE == 4)     // MFMA + 8 VALU
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3\n v_fma_f32 %6, %6, %2, %3\n v_fma_f32 %7, %7, %2, %3\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n v_fma_f32 %10, %10, %2, %3\n v_fma_f32 %11, %11, %2, %3\n"
                         "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3\n v_fma_f32 %6, %6, %2, %3\n v_fma_f32 %7, %7, %2, %3\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n v_fma_f32 %10, %10, %2, %3\n v_fma_f32 %11, %11, %2, %3\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        else if (MODE == 5)     // 8 VALU only (no MFMA): VALU issue cost
            asm volatile(".rept 32\n v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3\n v_fma_f32 %6, %6, %2, %3\n v_fma_f32 %7, %7, %2, %3\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n v_fma_f32 %10, %10, %2, %3\n v_fma_f32 %11, %11, %2, %3\n"
                         "v_fma_f32 %4, %4, %2, %3\n v_fma_f32 %5, %5, %2, %3\n v_fma_f32 %6, %6, %2, %3\n v_fma_f32 %7, %7, %2, %3\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n v_fma_f32 %10, %10, %2, %3\n v_fma_f32 %11, %11, %2, %3\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        else if (MODE == 6)     // MFMA + 1 transcendental + 2 VALU
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %2, %3\n v_fma_f32 %6, %6, %2, %3\n"
                         "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_exp_f32 %7, %7\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        else if (MODE == 7)     // MFMA + 1 ds_read_b32 + 2 VALU (operand of the NEXT mfma comes from LDS)
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n ds_read_b32 %12, %14\n v_fma_f32 %5, %5, %2, %3\n v_fma_f32 %6, %6, %2, %3\n"
                         "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n ds_read_b32 %13, %14 offset:256\n v_fma_f32 %8, %8, %2, %3\n v_fma_f32 %9, %9, %2, %3\n .endr\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(l0), "v"(l1), "v"((unsigned)(size_t)lp));
        else if (MODE == 8)     // MFMA + 4 f64 adds
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n"
                         "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"((double)x0), "v"((double)x1));
        else if (MODE == 9)     // dependent chain: ONE accumulator
            asm volatile(".rept 64\n v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv));
        else if (MODE == 10)    // MFMA whose A operand is produced by the VALU instruction right before it
            asm volatile(".rept 32\n v_mul_f32 %4, %2, %3\n v_mfma_f32_16x16x4_f32 %0, %4, %3, %0\n v_mul_f32 %5, %2, %3\n v_mfma_f32_16x16x4_f32 %1, %5, %3, %1\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(av), "v"(bv), "v"(x0), "v"(x1));

        else if (MODE == 11)    // bf16 XDL mfma only (K = 32), two chains
            asm volatile(".rept 32\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(A4), "v"(B4));
        else if (MODE == 12)    // bf16 mfma + 4 fma
            asm volatile(".rept 32\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n"
                         "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n v_fma_f32 %10, %10, %12, %13\n v_fma_f32 %11, %11, %12, %13\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(A4), "v"(B4), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(av), "v"(bv));
        else if (MODE == 13)    // bf16 mfma + 8 fma
            asm volatile(".rept 32\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n v_fma_f32 %10, %10, %12, %13\n v_fma_f32 %11, %11, %12, %13\n"
                         "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n v_fma_f32 %10, %10, %12, %13\n v_fma_f32 %11, %11, %12, %13\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(A4), "v"(B4), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(av), "v"(bv));
        else if (MODE == 14)    // f32 mfma and bf16 mfma alternating: do the two matrix paths overlap?
            asm volatile(".rept 32\n v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(A4), "v"(B4), "v"(av), "v"(bv));
        else if (MODE == 15)    // split of one f32 into three bf16 pieces, packed two at a time (per pair: 10 VALU)
            asm volatile(".rept 32\n"
                         "v_cvt_pk_bf16_f32 %6, %4, %5\n v_lshlrev_b32 %8, 16, %6\n v_and_b32 %9, 0xffff0000, %6\n v_sub_f32 %8, %4, %8\n v_sub_f32 %9, %5, %9\n"
                         "v_cvt_pk_bf16_f32 %7, %8, %9\n v_lshlrev_b32 %10, 16, %7\n v_and_b32 %11, 0xffff0000, %7\n v_sub_f32 %8, %8, %10\n v_sub_f32 %9, %9, %11\n"
                         "v_cvt_pk_bf16_f32 %10, %8, %9\n v_fma_f32 %4, %4, %12, %13\n"
                         "v_cvt_pk_bf16_f32 %6, %4, %5\n v_lshlrev_b32 %8, 16, %6\n v_and_b32 %9, 0xffff0000, %6\n v_sub_f32 %8, %4, %8\n v_sub_f32 %9, %5, %9\n"
                         "v_cvt_pk_bf16_f32 %7, %8, %9\n v_lshlrev_b32 %10, 16, %7\n v_and_b32 %11, 0xffff0000, %7\n v_sub_f32 %8, %8, %10\n v_sub_f32 %9, %9, %11\n"
                         "v_cvt_pk_bf16_f32 %10, %8, %9\n v_fma_f32 %5, %5, %12, %13\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(A4), "v"(B4), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(av), "v"(bv));
        else if (MODE == 16)    // bf16 mfma (K = 16 variant) only
            asm volatile(".rept 32\n v_mfma_f32_16x16x16_bf16 %0, %2, %3, %0\n v_mfma_f32_16x16x16_bf16 %1, %2, %3, %1\n .endr"
                         : "+v"(a0), "+v"(a1) : "v"(A2), "v"(B2));
        else if (MODE == 17)    // bf16 mfma + 2 ds_read_b128 + 4 fma
            asm volatile(".rept 32\n v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n ds_read_b128 %14, %16\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n"
                         "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n ds_read_b128 %15, %16 offset:4096\n v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n v_fma_f32 %10, %10, %12, %13\n v_fma_f32 %11, %11, %12, %13\n .endr\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1) : "v"(A4), "v"(B4), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(av), "v"(bv), "v"(L0), "v"(L1), "v"((unsigned)(size_t)(lds + 4 * (threadIdx.x & 63))));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + l0 + l1 + L0[0] + L1[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char *name, int threads, float *out, unsigned long long *cyc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(out, cyc, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, cyc, 0.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double slots = (double)OUTER * REP;
    printf("%-34s waves/SIMD=%d  memtime/slot=%7.2f  wall_ns/slot=%6.2f\n", name, threads / 256, c / slots, ms * 1e6 / slots);
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 64);
    for (int th : {256, 512}) {
        if (th == 256) {
#define ALL(T)                                                                                           \
    run<0>("mfma only (2 chains)", T, out, cyc); run<9>("mfma only (1 chain)", T, out, cyc);            \
    run<1>("mfma + 2 fma", T, out, cyc); run<2>("mfma + 4 fma", T, out, cyc);                            \
    run<3>("mfma + 6 fma", T, out, cyc); run<4>("mfma + 8 fma", T, out, cyc);                            \
    run<5>("8 fma only", T, out, cyc); run<6>("mfma + exp + 2 fma", T, out, cyc);                        \
    run<7>("mfma + ds_read + 2 fma", T, out, cyc); run<8>("mfma + 4 pk_fma", T, out, cyc);              \
    run<10>("mul -> mfma (A operand dep)", T, out, cyc);                                              \
    run<11>("bf16 k32 mfma only", T, out, cyc); run<16>("bf16 k16 mfma only", T, out, cyc);             \
    run<12>("bf16 k32 mfma + 4 fma", T, out, cyc); run<13>("bf16 k32 mfma + 8 fma", T, out, cyc);       \
    run<14>("f32 mfma + bf16 mfma pairs", T, out, cyc); run<15>("split 2 f32 -> 3 bf16 (x2, 24 valu)", T, out, cyc); \
    run<17>("bf16 k32 mfma + ds_read_b128 + 4 fma", T, out, cyc);
            ALL(256)
        } else {
            ALL(512)
        }
    }
    return 0;
}
