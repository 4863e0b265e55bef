// How does v_mfma_f32_16x16x32_bf16 round the sum  C + sum_k a_k b_k  into its float32 accumulator?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_round.hip -o tools/ubench/mfma_round && tools/ubench/mfma_round
// Each test starts from C = 1 and adds, n times, ONE product p (a in slot k = 0, b = 1) that is a fraction of ulp(1) = 2^-23:
// round-to-nearest-even gives 1 + n ulp for p = 0.75 ulp and 1 for p = 0.25 ulp; truncation gives 1 for both; the sign-symmetric
// case (C = -1, p < 0) tells toward-zero from toward-minus-infinity.  Also: K products of 0.25 ulp each inside ONE instruction
// (is the K-sum formed exactly before it meets C?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float c0, float pa, float pb, int nk, int n, float *out) {
    const int lane = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)0.f; b[j] = (__bf16)0.f; }
    // A[row = lane & 15][k = 8 (lane >> 4) + j], B[k = 8 (lane >> 4) + j][col = lane & 15]: fill the first nk slots of K
    for (int j = 0; j < 8; ++j) {
        const int kk = 8 * (lane >> 4) + j;
        if (kk < nk) { a[j] = (__bf16)pa; b[j] = (__bf16)pb; }
    }
    f32x4 c = {c0, c0, c0, c0};
    for (int i = 0; i < n; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}
int main() {
    float *d; hipMalloc(&d, 4);
    const float ulp = ldexpf(1.f, -23);
    struct { const char *name; float c0, pa, pb; int nk, n; } t[] = {
        {"C=+1, p=+0.75 ulp x 1000 (RNE: 1+1000 ulp, trunc: 1)", 1.f, 0.75f, ulp, 1, 1000},
        {"C=+1, p=+0.25 ulp x 1000 (RNE: 1, trunc: 1)", 1.f, 0.25f, ulp, 1, 1000},
        {"C=-1, p=-0.75 ulp x 1000", -1.f, -0.75f, ulp, 1, 1000},
        {"C=+1, p=-0.75 ulp(0.5) x 1000 (RNE: steps down)", 1.f, -0.75f, ulp / 2, 1, 1000},
        {"C=+1, 4 products of 0.25 ulp in one MFMA x 1000 (exact K-sum + RNE: 1+1000 ulp)", 1.f, 0.25f, ulp, 4, 1000},
        {"C=+1, 32 products of 0.0625 ulp in one MFMA x 1000 (sum 2 ulp: 1+2000 ulp)", 1.f, 0.0625f, ulp, 32, 1000},
        {"C=+1, 3 products of 0.25 ulp (sum 0.75) x 1000", 1.f, 0.25f, ulp, 3, 1000},
    };
    for (auto &x : t) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, x.c0, x.pa, x.pb, x.nk, x.n, d);
        float r; hipMemcpy(&r, d, 4, hipMemcpyDeviceToHost);
        printf("%-90s -> %+.10f = %+d ulp from C\n", x.name, r, (int)lrintf((r - x.c0) / ulp));
    }
    return 0;
}
