// Mean signed relative error (bias) and rms of the hardware v_rcp_f32 / v_exp_f32 / v_log_f32 against float64, over
// the argument ranges the QFA kernels use.  A bias of a fraction of an ulp is invisible in any per-element result, but
// the scalar gradients are sums of ~1e5..1e8 terms that cancel 50-900x while such a bias adds up coherently.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float *x, double *out, int n, int which) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    double ref, got;
    if (which == 0) { got = __builtin_amdgcn_rcpf(v); ref = 1.0 / (double)v; }
    else if (which == 1) { got = __builtin_amdgcn_exp2f(v); ref = exp2((double)v); }
    else if (which == 2) { got = __builtin_amdgcn_logf(v); ref = log2((double)v); }
    else if (which == 3) { got = 1.0f / v; ref = 1.0 / (double)v; }
    else { got = __builtin_amdgcn_rcpf(v); got = got * (2.0f - v * (float)got); ref = 1.0 / (double)v; }   // one Newton step
    out[i] = (got - ref) / ref;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    float *d; double *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 8);
    std::vector<double> r(n);
    const char *names[] = {"v_rcp_f32", "v_exp_f32", "v_log_f32", "IEEE 1/x", "rcp + Newton", "v_log_f32", "v_log_f32", "v_log_f32", "v_log_f32", "v_log_f32", "v_log_f32", "v_exp_f32", "v_exp_f32", "v_exp_f32"};
    const double lo[] = {0.01, -3.0, 1.0, 0.01, 0.01, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5, -1.0, 0.0, 1.0}, hi[] = {4.0, 3.0, 6.0, 4.0, 4.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 0.0, 1.0, 3.0};
    const int kind[] = {0, 1, 2, 3, 4, 2, 2, 2, 2, 2, 2, 1, 1, 1};
    for (int w = 0; w < 14; ++w) {
        unsigned s = 12345u + w;
        for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(lo[w] + (hi[w] - lo[w]) * (s >> 8) / 16777216.0); }
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        k<<<n / 256, 256>>>(d, o, n, kind[w]);
        hipMemcpy(r.data(), o, n * 8, hipMemcpyDeviceToHost);
        double m = 0, q = 0, mx = 0;
        for (int i = 0; i < n; ++i) { m += r[i]; q += r[i] * r[i]; mx = fmax(mx, fabs(r[i])); }
        printf("%-14s x in [%g, %g]: mean signed rel err %+.3e  rms %.3e  max %.3e  (2^-24 = 5.96e-08)\n", names[w], lo[w], hi[w], m / n, sqrt(q / n), mx);
    }
    return 0;
}
