// How fast does LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) move L2-RESIDENT data into LDS, per CU, and what
// does a request cost the issuing wave?  One workgroup of NWAVES waves per CU; every wave copies pieces of a 64-KiB region
// (re-read in a loop: L2 hits) into its own LDS area, `burst` pieces between waits.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/ldsdma_rate.hip -o tools/ubench/ldsdma_rate && tools/ubench/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void glds16a(const void *sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int BURST>
__global__ __launch_bounds__(512) void k_dma(const unsigned char *src, int iters, unsigned long long *cyc, float *sink) {
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char *base = src + (size_t)(blockIdx.x % 64) * 65536;
    const unsigned dst0 = (unsigned)(size_t)(lds + wv * BURST * 1024);
    unsigned long long t0, t1, tissue = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        unsigned long long a, b;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a)::"memory");
#pragma unroll
        for (int j = 0; j < BURST; ++j) {
            const unsigned piece = (unsigned)((it * BURST + j) * 8 + wv) & 63u;
            const unsigned char *p = base + piece * 1024;
            const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)p), hi = __builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)p >> 32));
            glds16a((const void *)(((unsigned long long)hi << 32) | lo), lane * 16u, dst0 + j * 1024);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b)::"memory");
        tissue += b - a;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0 && blockIdx.x == 7) { cyc[2 * wv] = t1 - t0; cyc[2 * wv + 1] = tissue; }
    if (lds[threadIdx.x] == 123) sink[0] = 1.f;
}
template <int BURST>
void run(int nwaves, const unsigned char *src, unsigned long long *dcyc, float *sink) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_dma<BURST><<<256, 64 * nwaves, nwaves * BURST * 1024, 0>>>(src, 10, dcyc, sink);
    hipEventRecord(e0);
    k_dma<BURST><<<256, 64 * nwaves, nwaves * BURST * 1024, 0>>>(src, iters, dcyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[16];
    hipMemcpy(c, dcyc, sizeof(c), hipMemcpyDeviceToHost);
    const double bytes = 256.0 * nwaves * BURST * 1024.0 * iters;
    printf("waves/CU %d  burst %d: %.2f TB/s chip, %.1f B/clk/CU (clock from s_memtime: %.0f cycles per burst), issue %.0f cycles per piece\n", nwaves, BURST,
           bytes / (ms * 1e-3) / 1e12, nwaves * BURST * 1024.0 / ((double)c[0] / iters), (double)c[0] / iters, (double)c[1] / iters / BURST);
}
int main() {
    unsigned char *src; unsigned long long *dcyc; float *sink;
    hipMalloc(&src, 64 * 65536); hipMemset(src, 1, 64 * 65536);
    hipMalloc(&dcyc, 16 * sizeof(unsigned long long)); hipMalloc(&sink, 4);
    for (int nw : {1, 2, 4, 8}) { run<1>(nw, src, dcyc, sink); run<4>(nw, src, dcyc, sink); run<8>(nw, src, dcyc, sink); }
    return 0;
}
