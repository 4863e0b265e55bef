// HBM write rate with the posterior writer's store pattern: a wave owns 16 rows of two row-major (B, Npix) float arrays
// and writes, per 32-pixel tile and array, 16 x 128 contiguous bytes as 8-byte stores (k_predict_x) -- against 16-byte
// stores over 64-pixel tiles (16 rows x 256 B) and a fully linear write.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/row_store.hip -o /tmp/row_store && /tmp/row_store
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W>      // W floats per lane and store: 2 (32-pixel tiles) or 4 (64-pixel tiles)
__global__ __launch_bounds__(256, 2) void k_rows(float *a0, float *a1, int B, int Npix, int nblk_per_wg) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lo = lane & 15, g = lane >> 4;
    const int ntiles = Npix / (16 * W);
    for (int blk = blockIdx.x; blk * 64 < B; blk += gridDim.x) {
        const int s0 = blk * 64 + wv * 16;
        for (int t = 0; t < ntiles; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = s0 + 4 * g + r;
                if (s < B) {
                    const size_t o = (size_t)s * Npix + 16 * W * t + W * lo;
                    if (W == 2) {
                        *reinterpret_cast<f2 *>(a0 + o) = f2{(float)t, (float)r};
                        *reinterpret_cast<f2 *>(a1 + o) = f2{(float)r, (float)t};
                    } else {
                        *reinterpret_cast<f4 *>(a0 + o) = f4{(float)t, (float)r, 1.f, 2.f};
                        *reinterpret_cast<f4 *>(a1 + o) = f4{(float)r, (float)t, 3.f, 4.f};
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void k_linear(f4 *a0, f4 *a1, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        a0[i] = f4{1.f, 2.f, 3.f, 4.f};
        a1[i] = f4{4.f, 3.f, 2.f, 1.f};
    }
}

int main() {
    const int B = 100000, Npix = 4000;
    const size_t n = (size_t)B * Npix;
    float *d0, *d1;
    hipMalloc(&d0, n * 4);
    hipMalloc(&d1, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double bytes = (double)n * 8;
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
        printf("%-52s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    time("linear 16-byte stores, 2048 workgroups", [&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (f4 *)d0, (f4 *)d1, n / 4); });
    time("linear 16-byte stores, 512 workgroups", [&] { hipLaunchKernelGGL(k_linear, dim3(512), dim3(256), 0, 0, (f4 *)d0, (f4 *)d1, n / 4); });
    time("16 rows x 128 B per wave, 8-byte stores", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(512), dim3(256), 0, 0, d0, d1, B, Npix, 0); });
    time("16 rows x 256 B per wave, 16-byte stores", [&] { hipLaunchKernelGGL(k_rows<4>, dim3(512), dim3(256), 0, 0, d0, d1, B, Npix, 0); });
    time("16 rows x 128 B, 8-byte stores, 1563 workgroups", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(1563), dim3(256), 0, 0, d0, d1, B, Npix, 0); });
    return 0;
}
