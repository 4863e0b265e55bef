// How fast can HBM be streamed with pass 1's access pattern?  A wave owns 16 rows of a row-major (B, Npix) float array
// and walks the pixel axis in 32-pixel tiles: per tile and array 16 x 128 contiguous bytes, rows Npix * 4 bytes apart,
// D tiles ahead in registers -- against a fully linear read of the same bytes.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/row_stream.hip -o /tmp/row_stream && /tmp/row_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int D, int NARR>
__global__ __launch_bounds__(256, 2) void k_rows(const float *const *arr, int B, int Npix, float *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sl = lane & 15, g = lane >> 4;
    const int ntiles = Npix / 32;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int blk = blockIdx.x; blk * 64 < B; blk += gridDim.x) {
        const int s = blk * 64 + wv * 16 + sl;
        if (s >= B) continue;
        f4 buf[D][NARR][2];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int a = 0; a < NARR; ++a) {
                const float *p = arr[a] + (size_t)s * Npix + 32 * d + 8 * g;
                buf[d][a][0] = *reinterpret_cast<const f4 *>(p);
                buf[d][a][1] = *reinterpret_cast<const f4 *>(p + 4);
            }
        for (int t = 0; t < ntiles; t += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int a = 0; a < NARR; ++a) {
                    acc += buf[d][a][0] * buf[d][a][1];
                    const int tn = t + d + D;
                    if (tn < ntiles) {
                        const float *p = arr[a] + (size_t)s * Npix + 32 * tn + 8 * g;
                        buf[d][a][0] = *reinterpret_cast<const f4 *>(p);
                        buf[d][a][1] = *reinterpret_cast<const f4 *>(p + 4);
                    }
                }
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

// 64-pixel tiles: a wave = 8 rows x 256 contiguous bytes per instruction pair... here: lane (sl = lane & 7, g = lane >> 3)
template <int D, int NARR>
__global__ __launch_bounds__(256, 2) void k_rows256(const float *const *arr, int B, int Npix, float *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sl = lane & 7, g = lane >> 3;
    const int ntiles = Npix / 64;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int blk = blockIdx.x; blk * 32 < B; blk += gridDim.x) {
        const int s = blk * 32 + wv * 8 + sl;
        if (s >= B) continue;
        f4 buf[D][NARR][2];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int a = 0; a < NARR; ++a) {
                const float *p = arr[a] + (size_t)s * Npix + 64 * d + 8 * g;
                buf[d][a][0] = *reinterpret_cast<const f4 *>(p);
                buf[d][a][1] = *reinterpret_cast<const f4 *>(p + 4);
            }
        for (int t = 0; t < ntiles; t += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int a = 0; a < NARR; ++a) {
                    acc += buf[d][a][0] * buf[d][a][1];
                    const int tn = t + d + D;
                    if (tn < ntiles) {
                        const float *p = arr[a] + (size_t)s * Npix + 64 * tn + 8 * g;
                        buf[d][a][0] = *reinterpret_cast<const f4 *>(p);
                        buf[d][a][1] = *reinterpret_cast<const f4 *>(p + 4);
                    }
                }
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

__global__ __launch_bounds__(256, 2) void k_linear(const float *const *arr, int narr, size_t n4, float *out) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    for (int a = 0; a < narr; ++a) {
        const f4 *p = reinterpret_cast<const f4 *>(arr[a]);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
            f4 v0 = p[i], v1 = i + stride < n4 ? p[i + stride] : acc, v2 = i + 2 * stride < n4 ? p[i + 2 * stride] : acc,
               v3 = i + 3 * stride < n4 ? p[i + 3 * stride] : acc;
            acc += v0 * v1 + v2 * v3;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

int main() {
    const int B = 100000, Npix = 4000, NARR = 2;
    const size_t n = (size_t)B * Npix;
    float *d[NARR], *out;
    const float **darr;
    for (int a = 0; a < NARR; ++a) {
        hipMalloc(&d[a], n * 4);
        hipMemset(d[a], 0, n * 4);
    }
    hipMalloc(&out, 4);
    hipMalloc(&darr, sizeof(float *) * NARR);
    hipMemcpy(darr, d, sizeof(float *) * NARR, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = (double)n * 4 * NARR;
    auto time = [&](const char *name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    time("linear, 512 workgroups", [&] { hipLaunchKernelGGL(k_linear, dim3(512), dim3(256), 0, 0, darr, NARR, n / 4, out); });
    time("linear, 2048 workgroups", [&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, darr, NARR, n / 4, out); });
    time("16 rows x 128 B per wave, 1 tile ahead", [&] { hipLaunchKernelGGL((k_rows<1, NARR>), dim3(512), dim3(256), 0, 0, darr, B, Npix, out); });
    time("16 rows x 128 B per wave, 2 tiles ahead", [&] { hipLaunchKernelGGL((k_rows<2, NARR>), dim3(512), dim3(256), 0, 0, darr, B, Npix, out); });
    time("16 rows x 128 B per wave, 4 tiles ahead", [&] { hipLaunchKernelGGL((k_rows<4, NARR>), dim3(512), dim3(256), 0, 0, darr, B, Npix, out); });
    time("16 rows x 128 B per wave, 8 tiles ahead", [&] { hipLaunchKernelGGL((k_rows<8, NARR>), dim3(512), dim3(256), 0, 0, darr, B, Npix, out); });
    time("8 rows x 256 B per wave, 1 tile ahead", [&] { hipLaunchKernelGGL((k_rows256<1, NARR>), dim3(512), dim3(256), 0, 0, darr, B, Npix, out); });
    time("8 rows x 256 B per wave, 2 tiles ahead", [&] { hipLaunchKernelGGL((k_rows256<2, NARR>), dim3(512), dim3(256), 0, 0, darr, B, Npix, out); });
    time("8 rows x 256 B per wave, 4 tiles ahead", [&] { hipLaunchKernelGGL((k_rows256<4, NARR>), dim3(512), dim3(256), 0, 0, darr, B, Npix, out); });
    return 0;
}
