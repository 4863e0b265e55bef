// Does an MFMA whose destination overlaps its A (or B) operand return the same result as one with a separate
// destination?  hipcc allocates such overlaps for the 4-register-destination MFMAs (k_grads_s3, round 2).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

// A in v[A0:A0+1], B in v[20:21], destination v[D0:D0+3]; registers 0..23 are clobbered
#define CASE16(NAME, D0, D1, D2, D3, DLO, DHI, ALO, AHI)                                                            \
    __device__ void NAME(unsigned a0, unsigned a1, unsigned b0, unsigned b1, float *o) {                            \
        float r0, r1, r2, r3;                                                                                       \
        asm volatile("v_mov_b32 v" #ALO ", %4\n v_mov_b32 v" #AHI ", %5\n v_mov_b32 v20, %6\n v_mov_b32 v21, %7\n"  \
                     "s_nop 7\n"                                                                                     \
                     "v_mfma_f32_16x16x16_bf16 v[" #DLO ":" #DHI "], v[" #ALO ":" #AHI "], v[20:21], 0\n"            \
                     "s_nop 15\n s_nop 15\n"                                                                         \
                     "v_mov_b32 %0, v" #D0 "\n v_mov_b32 %1, v" #D1 "\n v_mov_b32 %2, v" #D2 "\n v_mov_b32 %3, v" #D3 \
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)                                                   \
                     : "v"(a0), "v"(a1), "v"(b0), "v"(b1)                                                           \
                     : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",      \
                       "v20", "v21");                                                                               \
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;                                                                 \
    }
// B in v[BLO:BHI], A in v[20:21]
#define CASE16B(NAME, D0, D1, D2, D3, DLO, DHI, BLO, BHI)                                                           \
    __device__ void NAME(unsigned a0, unsigned a1, unsigned b0, unsigned b1, float *o) {                            \
        float r0, r1, r2, r3;                                                                                       \
        asm volatile("v_mov_b32 v" #BLO ", %6\n v_mov_b32 v" #BHI ", %7\n v_mov_b32 v20, %4\n v_mov_b32 v21, %5\n"  \
                     "s_nop 7\n"                                                                                     \
                     "v_mfma_f32_16x16x16_bf16 v[" #DLO ":" #DHI "], v[20:21], v[" #BLO ":" #BHI "], 0\n"            \
                     "s_nop 15\n s_nop 15\n"                                                                         \
                     "v_mov_b32 %0, v" #D0 "\n v_mov_b32 %1, v" #D1 "\n v_mov_b32 %2, v" #D2 "\n v_mov_b32 %3, v" #D3 \
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)                                                   \
                     : "v"(a0), "v"(a1), "v"(b0), "v"(b1)                                                           \
                     : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",      \
                       "v20", "v21");                                                                               \
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;                                                                 \
    }

CASE16(ref_sep, 0, 1, 2, 3, 0, 3, 8, 9)        // separate
CASE16(a_hi, 0, 1, 2, 3, 0, 3, 2, 3)           // A = upper half of the destination (the k_grads_s3 case)
CASE16(a_lo, 0, 1, 2, 3, 0, 3, 0, 1)           // A = lower half
CASE16(a_mid, 2, 3, 4, 5, 2, 5, 2, 3)          // destination aligned to 2 only
CASE16(a_edge, 2, 3, 4, 5, 2, 5, 4, 5)
CASE16B(b_hi, 0, 1, 2, 3, 0, 3, 2, 3)
CASE16B(b_lo, 0, 1, 2, 3, 0, 3, 0, 1)
CASE16B(b_mid, 2, 3, 4, 5, 2, 5, 4, 5)

__global__ void k(const unsigned *in, float *out) {
    const int l = threadIdx.x;
    const unsigned a0 = in[l * 4], a1 = in[l * 4 + 1], b0 = in[l * 4 + 2], b1 = in[l * 4 + 3];
    float *o = out + l * 4;
    ref_sep(a0, a1, b0, b1, o);
    a_hi(a0, a1, b0, b1, o + 256);
    a_lo(a0, a1, b0, b1, o + 512);
    a_mid(a0, a1, b0, b1, o + 768);
    a_edge(a0, a1, b0, b1, o + 1024);
    b_hi(a0, a1, b0, b1, o + 1280);
    b_lo(a0, a1, b0, b1, o + 1536);
    b_mid(a0, a1, b0, b1, o + 1792);
}

// the A operand's second register is written by v_cvt_pk_bf16_f32 NOPS wait states in front of the MFMA
#define CASECVT(NAME, D0, D1, D2, D3, DLO, DHI, ALO, AHI, NOPS)                                                     \
    __device__ void NAME(unsigned a0, float x, float y, unsigned b0, unsigned b1, float *o) {                       \
        float r0, r1, r2, r3;                                                                                       \
        asm volatile("v_mov_b32 v" #ALO ", %4\n v_mov_b32 v20, %7\n v_mov_b32 v21, %8\n v_mov_b32 v22, %5\n"        \
                     "v_mov_b32 v23, %6\n s_nop 7\n"                                                                 \
                     "v_cvt_pk_bf16_f32 v" #AHI ", v22, v23\n" NOPS                                                  \
                     "v_mfma_f32_16x16x16_bf16 v[" #DLO ":" #DHI "], v[" #ALO ":" #AHI "], v[20:21], 0\n"            \
                     "s_nop 15\n s_nop 15\n"                                                                         \
                     "v_mov_b32 %0, v" #D0 "\n v_mov_b32 %1, v" #D1 "\n v_mov_b32 %2, v" #D2 "\n v_mov_b32 %3, v" #D3 \
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)                                                   \
                     : "v"(a0), "v"(x), "v"(y), "v"(b0), "v"(b1)                                                    \
                     : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13",      \
                       "v20", "v21", "v22", "v23", "v30", "v31");                                                                 \
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;                                                                 \
    }
CASECVT(cvt_sep_7, 0, 1, 2, 3, 0, 3, 8, 9, "s_nop 7\n")
CASECVT(cvt_sep_n, 0, 1, 2, 3, 0, 3, 8, 9, "")
CASECVT(cvt_ovl_n, 0, 1, 2, 3, 0, 3, 2, 3, "")
CASECVT(cvt_ovl_0, 0, 1, 2, 3, 0, 3, 2, 3, "s_nop 0\n")
CASECVT(cvt_ovl_1, 0, 1, 2, 3, 0, 3, 2, 3, "s_nop 1\n")
CASECVT(cvt_ovl_2, 0, 1, 2, 3, 0, 3, 2, 3, "s_nop 2\n")
CASECVT(cvt_ovl_3, 0, 1, 2, 3, 0, 3, 2, 3, "s_nop 3\n")
CASECVT(cvt_ovl_4, 0, 1, 2, 3, 0, 3, 2, 3, "s_nop 4\n")
CASECVT(cvt_ovl_5, 0, 1, 2, 3, 0, 3, 2, 3, "s_nop 5\n")
CASECVT(cvt_lo_n, 0, 1, 2, 3, 0, 3, 0, 1, "")
CASECVT(cvt_v_n, 0, 1, 2, 3, 0, 3, 2, 3, "v_mov_b32 v30, v31\n v_mov_b32 v30, v31\n v_mov_b32 v30, v31\n")

__global__ void k2(const unsigned *in, const float *xy, float *out) {
    const int l = threadIdx.x;
    const unsigned a0 = in[l * 4], b0 = in[l * 4 + 2], b1 = in[l * 4 + 3];
    const float x = xy[2 * l], y = xy[2 * l + 1];
    float *o = out + l * 4;
    cvt_sep_7(a0, x, y, b0, b1, o);
    cvt_sep_n(a0, x, y, b0, b1, o + 256);
    cvt_ovl_n(a0, x, y, b0, b1, o + 512);
    cvt_ovl_0(a0, x, y, b0, b1, o + 768);
    cvt_ovl_1(a0, x, y, b0, b1, o + 1024);
    cvt_ovl_2(a0, x, y, b0, b1, o + 1280);
    cvt_ovl_3(a0, x, y, b0, b1, o + 1536);
    cvt_ovl_4(a0, x, y, b0, b1, o + 1792);
    cvt_ovl_5(a0, x, y, b0, b1, o + 2048);
    cvt_lo_n(a0, x, y, b0, b1, o + 2304);
    cvt_v_n(a0, x, y, b0, b1, o + 2560);
}

static unsigned short bf(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    return (unsigned short)(u >> 16);
}

int main() {
    unsigned h[256];
    srand(7);
    for (int i = 0; i < 256; ++i) {
        float x = (float)rand() / RAND_MAX * 2.f - 1.f, y = (float)rand() / RAND_MAX * 2.f - 1.f;
        h[i] = bf(x) | ((unsigned)bf(y) << 16);
    }
    unsigned *din;
    float *dout, ho[4096];
    hipMalloc(&din, sizeof h);
    hipMalloc(&dout, sizeof ho);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    const char *nm[8] = {"separate", "A = D[2:3]", "A = D[0:1]", "A=D[0:1] D@2", "A=D[2:3] D@2", "B = D[2:3]", "B = D[0:1]", "B=D[2:3] D@2"};
    for (int c = 0; c < 8; ++c) {
        int bad = 0;
        double worst = 0;
        for (int i = 0; i < 256; ++i) {
            double d = fabs((double)ho[c * 256 + i] - ho[i]);
            if (d != 0) ++bad;
            if (d > worst) worst = d;
        }
        printf("%-12s  differing entries %3d of 256, max |diff| %.3g\n", nm[c], bad, worst);
    }
    float hxy[128], *dxy;
    for (int i = 0; i < 128; ++i) hxy[i] = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    hipMalloc(&dxy, sizeof hxy);
    hipMemcpy(dxy, hxy, sizeof hxy, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, din, dxy, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    const char *nm2[11] = {"cvt, s_nop 7, separate", "cvt, no nop, separate", "cvt, no nop, A = D[2:3]",
                           "cvt, s_nop 0, A = D[2:3]", "cvt, s_nop 1, A = D[2:3]", "cvt, s_nop 2, A = D[2:3]",
                           "cvt, s_nop 3, A = D[2:3]", "cvt, s_nop 4, A = D[2:3]", "cvt, s_nop 5, A = D[2:3]",
                           "cvt, no nop, A = D[0:1]", "cvt, 3 v_mov, A = D[2:3]"};
    for (int c = 0; c < 11; ++c) {
        int bad = 0;
        double worst = 0;
        for (int i = 0; i < 256; ++i) {
            double d = fabs((double)ho[c * 256 + i] - ho[i]);
            if (d != 0) ++bad;
            if (d > worst) worst = d;
        }
        printf("%-26s differing entries %3d of 256, max |diff| %.3g\n", nm2[c], bad, worst);
    }
    return 0;
}
