// LDS-DMA source alignment on gfx950 (the pass-2 kernel streams spectra rows whose starts are only 4-byte aligned, and
// mask rows with no alignment at all, straight into LDS):
//   1. global_load_lds_dwordx4 (16 bytes per lane) from src + shift + 16 * lane, shift = 0, 4, 8, 12;
//   2. global_load_lds_dword (4 bytes per lane) from src + shift + 4 * lane, shift = 0..3 (byte-unaligned sources).
// Each block copies through LDS and writes LDS out; the host compares with a byte-wise copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
template <int BYTES>
__global__ void k(const unsigned char *src, int shift, unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned char *g = src + shift + BYTES * threadIdx.x;
    if (BYTES == 16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)lds, 4, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned char> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (unsigned char)(i * 37 + (i >> 8));
    unsigned char *d; unsigned *o;
    hipMalloc(&d, 8192); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode)
        for (int shift = 0; shift < (mode ? 4 : 16); shift += (mode ? 1 : 4)) {
            hipMemset(o, 0, 1024);
            const int nbytes = mode ? 256 : 1024;
            if (mode) k<4><<<1, 64>>>(d, shift, o); else k<16><<<1, 64>>>(d, shift, o);
            std::vector<unsigned char> r(1024);
            hipError_t e = hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < nbytes; ++i) bad += r[i] != h[i + shift];
            printf("%s shift %2d: err %d, %d of %d bytes wrong\n", mode ? "dword  " : "dwordx4", shift, (int)e, bad, nbytes);
        }
    return 0;
}
