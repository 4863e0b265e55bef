// Numerics check: 16x16 tile of sum_k A[m][k] B[k][n] accumulated over K, three ways:
//   (a) v_mfma_f32_16x16x4_f32 (exact f32 fma chain), (b) operands split into three bf16 pieces (RNE), six
//   v_mfma_f32_16x16x32_bf16 per K-step of 32 (hh, hm, mh, hl, lh, mm; small terms first), (c) the 3-term
//   variant (hh, hm, mh), (d) round 5: operands split into TWO f16 pieces (11 + 11 bits; rows of A and columns of B scaled by a power
//   of two so that their largest element is in [2^13, 2^14): f16 has 5 exponent bits), three v_mfma_f32_16x16x32_f16 (hm, mh, hh) and (e)
//   four (+ mm).  Reference: float64 on the host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h(float x0, float x1, unsigned &h, unsigned &m) {          // two f16 pieces, round to nearest
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 m0 = (_Float16)(x0 - (float)h0), m1 = (_Float16)(x1 - (float)h1);
    h = __builtin_bit_cast(unsigned, f16x2{h0, h1});
    m = __builtin_bit_cast(unsigned, f16x2{m0, m1});
}
__device__ __forceinline__ f32x4 xdlh(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}
__device__ __forceinline__ f32x4 xdl(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ void k(const float *A, const float *B, int K, float *out) {     // A [16][K], B [K][16]
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f32x4 c32 = {0, 0, 0, 0}, c6 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, h3 = {0, 0, 0, 0}, h4 = {0, 0, 0, 0};
    float amax = 0.f, bmax = 0.f;                                 // row r of A, column r of B
    for (int kk = 0; kk < K; ++kk) { amax = fmaxf(amax, fabsf(A[r * K + kk])); bmax = fmaxf(bmax, fabsf(B[kk * 16 + r])); }
    int ea, eb;
    frexpf(amax, &ea); frexpf(bmax, &eb);
    const float sa = ldexpf(1.f, 14 - ea), sb = ldexpf(1.f, 14 - eb);      // largest element in [2^13, 2^14), as the product path does
    for (int k0 = 0; k0 < K; k0 += 4) c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k0 + g], B[(k0 + g) * 16 + r], c32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        u32x4 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 4; ++j) {
            unsigned h, m, l;
            split2(A[r * K + k0 + 8 * g + 2 * j], A[r * K + k0 + 8 * g + 2 * j + 1], h, m, l);
            ah[j] = h; am[j] = m; al[j] = l;
            split2(B[(k0 + 8 * g + 2 * j) * 16 + r], B[(k0 + 8 * g + 2 * j + 1) * 16 + r], h, m, l);
            bh[j] = h; bm[j] = m; bl[j] = l;
        }
        c6 = xdl(al, bh, c6); c6 = xdl(ah, bl, c6); c6 = xdl(am, bm, c6);
        c6 = xdl(am, bh, c6); c6 = xdl(ah, bm, c6); c6 = xdl(ah, bh, c6);
        c3 = xdl(am, bh, c3); c3 = xdl(ah, bm, c3); c3 = xdl(ah, bh, c3);
        u32x4 fah, fam, fbh, fbm;
        for (int j = 0; j < 4; ++j) {
            unsigned h, m;
            split2h(sa * A[r * K + k0 + 8 * g + 2 * j], sa * A[r * K + k0 + 8 * g + 2 * j + 1], h, m);
            fah[j] = h; fam[j] = m;
            split2h(sb * B[(k0 + 8 * g + 2 * j) * 16 + r], sb * B[(k0 + 8 * g + 2 * j + 1) * 16 + r], h, m);
            fbh[j] = h; fbm[j] = m;
        }
        h3 = xdlh(fam, fbh, h3); h3 = xdlh(fah, fbm, h3); h3 = xdlh(fah, fbh, h3);
        h4 = xdlh(fam, fbm, h4); h4 = xdlh(fam, fbh, h4); h4 = xdlh(fah, fbm, h4); h4 = xdlh(fah, fbh, h4);
    }
    // un-scale: element (row 4 g + i of A, column r of B): the row scale of A sits in lane (row), fetch it from there
    for (int i = 0; i < 4; ++i) {
        const float sra = __shfl(sa, 4 * g + i, 64);
        h3[i] = h3[i] / (sra * sb);
        h4[i] = h4[i] / (sra * sb);
    }
    for (int i = 0; i < 4; ++i) {
        out[(4 * g + i) * 16 + r] = c32[i];
        out[256 + (4 * g + i) * 16 + r] = c6[i];
        out[512 + (4 * g + i) * 16 + r] = c3[i];
        out[768 + (4 * g + i) * 16 + r] = h3[i];
        out[1024 + (4 * g + i) * 16 + r] = h4[i];
    }
}

static float f16_round(float x) { return (float)(_Float16)x; }
// x as the two float16 pieces of the kernel hold it (scale s): exact afterwards
static float quant2(float x, float s) { const float y = x * s, h = f16_round(y), m = f16_round(y - h); return (h + m) / s; }
int main(int argc, char **argv) {
    const bool quant = argc > 1 && argv[1][0] == 'q';       // "q": inputs pre-rounded to two float16 pieces -- what is left is the MFMA's own error and mm
    if (quant) printf("inputs quantised to two float16 pieces (row / column scales as in the kernel): the float16 columns show the MFMA's own error\n");
    std::mt19937 rng(7);
    for (int K : {32, 512, 4096}) {
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<float> A(16 * K), B(K * 16);
            std::normal_distribution<float> nd(0.f, 1.f);
            std::uniform_real_distribution<float> ud(0.f, 1.f);
            for (auto &x : A) x = mode == 0 ? nd(rng) : (mode == 1 ? ud(rng) : std::exp((mode == 2 ? 6.f : 2.f) * nd(rng)));   // signed / positive / wide range (2: e^(6 N), 3: e^(2 N))
            for (auto &x : B) x = mode == 1 ? ud(rng) : nd(rng) * (mode == 2 ? std::exp(3.f * nd(rng)) : (mode == 3 ? std::exp(1.5f * nd(rng)) : 1.f));
            if (quant) {
                for (int m = 0; m < 16; ++m) {
                    float mx = 0.f; for (int kk = 0; kk < K; ++kk) mx = std::max(mx, std::fabs(A[m * K + kk]));
                    int e; std::frexp(mx, &e); const float sc = std::ldexp(1.f, 14 - e);
                    for (int kk = 0; kk < K; ++kk) A[m * K + kk] = quant2(A[m * K + kk], sc);
                }
                for (int n = 0; n < 16; ++n) {
                    float mx = 0.f; for (int kk = 0; kk < K; ++kk) mx = std::max(mx, std::fabs(B[kk * 16 + n]));
                    int e; std::frexp(mx, &e); const float sc = std::ldexp(1.f, 14 - e);
                    for (int kk = 0; kk < K; ++kk) B[kk * 16 + n] = quant2(B[kk * 16 + n], sc);
                }
            }
            float *dA, *dB, *dO;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 1280 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            k<<<1, 64>>>(dA, dB, K, dO);
            std::vector<float> O(1280);
            hipMemcpy(O.data(), dO, 1280 * 4, hipMemcpyDeviceToHost);
            double e[5] = {0, 0, 0, 0, 0}, emax[5] = {0, 0, 0, 0, 0};
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) {
                    double ref = 0, mag = 0;
                    for (int kk = 0; kk < K; ++kk) { ref += (double)A[m * K + kk] * B[kk * 16 + n]; mag += std::fabs((double)A[m * K + kk] * B[kk * 16 + n]); }
                    for (int v = 0; v < 5; ++v) {
                        const double err = std::fabs(O[256 * v + m * 16 + n] - ref) / mag;
                        e[v] += err * err / 256; emax[v] = std::max(emax[v], err);
                    }
                }
            printf("K=%5d mode=%d  err/sum|ab|  f32 mfma: rms %.2e max %.2e | bf16x6: rms %.2e max %.2e | bf16x3: rms %.2e max %.2e | f16 2 pieces x3: rms %.2e max %.2e | x4: rms %.2e max %.2e\n", K, mode,
                   std::sqrt(e[0]), emax[0], std::sqrt(e[1]), emax[1], std::sqrt(e[2]), emax[2], std::sqrt(e[3]), emax[3], std::sqrt(e[4]), emax[4]);
            hipFree(dA); hipFree(dB); hipFree(dO);
        }
    }
    return 0;
}
