// What does the XDL pipe deliver on THIS chip under its power cap?  (round 5: the training step runs at 1 393 W of 1 400 W with the
// shader clock at ~1.9 GHz, profiles/r5_power_clocks.txt.)  Whole-chip loops of v_mfma_f32_16x16x32_bf16, two waves per SIMD, every
// mode run back to back for ~2.5 s; reports TFLOP/s and the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/xdl_power.hip -o tools/ubench/xdl_power && tools/ubench/xdl_power
// modes: 0 zero operands, registers | 1 random bf16 operands, registers | 2 random "piece" operands (A cycles through the hi / mid /
// lo pieces of random floats, as the six piece products of the product path do) | 3 as 2 + one ds_read_b128 per MFMA feeding the next
// B operand | 4 as 3 + two v_fma_f32 per MFMA | 5..8 as 4 with 8 / 24 / 48 / 80 idle issue cycles
// (s_nop) per MFMA and wave: the stalls of a real kernel -- how much of the idle time does the clock give back?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const u32x4 *__restrict__ in, float *out, unsigned long long *ticks, int reps) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 a[3], b[3];
    for (int j = 0; j < 3; ++j) {
        a[j] = MODE == 0 ? u32x4{0, 0, 0, 0} : in[(MODE == 1 ? 0 : 64 * (1 + j)) + lane + 256 * j * (MODE == 1)];
        b[j] = MODE == 0 ? u32x4{0, 0, 0, 0} : in[1024 + (MODE == 1 ? 0 : 64 * (1 + j)) + lane + 256 * j * (MODE == 1)];
    }
    for (int j = 0; j < 4; ++j)
        reinterpret_cast<u32x4 *>(lds + wave * 4096)[j * 64 + lane] = MODE == 0 ? u32x4{0, 0, 0, 0} : in[2048 + 64 * j + lane];
    __syncthreads();
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = 1.f + lane, x1 = 2.f + lane;
    const float p = 0.999f, q = 0.001f;
    const unsigned la = (unsigned)(size_t)(lds + wave * 4096 + lane * 16);
    u32x4 d0 = b[0], d1 = b[1];
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#define M(C, A, B) "v_mfma_f32_16x16x32_bf16 " C ", " A ", " B ", " C "\n"
#define SIX M("%0", "%4", "%7") M("%1", "%4", "%8") M("%2", "%5", "%7") M("%3", "%4", "%9") M("%0", "%6", "%7") M("%1", "%5", "%8")
#define SIXA M("%0", "%4", "%7") M("%1", "%4", "%8") M("%2", "%4", "%9") M("%3", "%5", "%7") M("%0", "%5", "%8") M("%1", "%6", "%7")
#define SIXB M("%0", "%4", "%7") M("%1", "%5", "%8") M("%2", "%6", "%7") M("%3", "%4", "%9") M("%0", "%5", "%7") M("%1", "%4", "%8")
// LDS modes: %0-3 accumulators, %4 %5 the B operands the ds_reads refill, %6 %7 the fma chains, %8-10 A pieces, %11 LDS address, %12 %13 p q
#define SIXL(X) M("%0", "%8", "%4") X "ds_read_b128 %4, %11\n" M("%1", "%8", "%5") X "ds_read_b128 %5, %11 offset:1024\n" \
                M("%2", "%9", "%4") X "ds_read_b128 %4, %11 offset:2048\n" M("%3", "%10", "%5") X "ds_read_b128 %5, %11 offset:3072\n" \
                M("%0", "%9", "%4") X "ds_read_b128 %4, %11\n" M("%1", "%10", "%5") X "ds_read_b128 %5, %11 offset:1024\n"
#define FMA2 "v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n"
#define LDSRUN(X)                                                                                                               \
    asm volatile(".rept 16\n" SIXL(X) ".endr\n s_waitcnt lgkmcnt(0)\n"                                                          \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(x0), "+v"(x1)                                  \
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(la), "v"(p), "v"(q)                                                         \
                 : "memory")
    for (int r = 0; r < reps; ++r) {
        if (MODE <= 2) {
            asm volatile(".rept 16\n" SIX ".endr\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]));
        } else if (MODE == 9) {
            asm volatile(".rept 16\n" SIXA ".endr\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]));
        } else if (MODE == 10) {
            asm volatile(".rept 16\n" SIXB ".endr\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]));
        } else if (MODE == 3) {
            LDSRUN("");
        } else if (MODE == 4) {
            LDSRUN(FMA2);
        } else if (MODE == 5) {
            LDSRUN(FMA2 "s_nop 7\n");
        } else if (MODE == 6) {
            LDSRUN(FMA2 "s_nop 7\n s_nop 7\n s_nop 7\n");
        } else if (MODE == 7) {
            LDSRUN(FMA2 "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n");
        } else {
            LDSRUN(FMA2 "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n");
        }
    }
    asm volatile("s_nop 7\n s_nop 7\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (lane == 0 && wave == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + __uint_as_float(d0[0] ^ d1[1]);
}

static unsigned short bf16(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float f32(unsigned short h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; }

template <int MODE>
void run(const u32x4 *in, float *out, unsigned long long *ticks, const char *what) {
    const int blocks = 1024, reps = MODE >= 6 ? 300 : 600;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) k<MODE><<<blocks, 512>>>(in, out, ticks, reps);
    hipDeviceSynchronize();
    float ms = 0; int n = 0; double total = 0;
    while (total < 2500.0) {                                     // ~2.5 s of back-to-back launches, the last 20 timed
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) k<MODE><<<blocks, 512>>>(in, out, ticks, reps);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        total += ms; ++n;
    }
    std::vector<unsigned long long> t(2 * blocks);
    hipMemcpy(t.data(), ticks, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> clk(blocks);
    for (int i = 0; i < blocks; ++i) clk[i] = (double)t[2 * i] / (double)t[2 * i + 1] * 100.0;
    std::sort(clk.begin(), clk.end());
    const double cyc = 0; (void)cyc;
    const double mf = (double)blocks * 8 * reps * 16 * 6;      // MFMAs per launch
    const double tf = mf * 16384.0 / (ms / 20 * 1e-3) * 1e-12;
    printf("mode %d  %-58s %8.1f TFLOP/s   clock %5.0f MHz   (%.3f ms per launch, %d x 20 launches)\n", MODE, what, tf, clk[blocks / 2], ms / 20, n);
    fflush(stdout);
}

int main() {
    std::vector<unsigned> h(4096 * 4);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    // u32x4 index 0..1023: A operands, 1024..2047: B, 2048..: LDS fill.  [0..63]: plain random bf16 (mode 1: + 256 j);
    // [64 (1 + j) ..]: piece j of random floats
    for (int side = 0; side < 3; ++side)
        for (int i = 0; i < 1024 * 4; ++i) {
            const int v = i / 4, e = i % 4;
            unsigned w = 0;
            for (int half = 0; half < 2; ++half) {
                unsigned short bits;
                const float x = rnd();
                const int blk = v / 64;
                if (side < 2 && blk >= 1 && blk <= 3) {       // piece blk - 1 of x
                    float rest = x; unsigned short pc = 0;
                    for (int j = 0; j <= blk - 1; ++j) { pc = bf16(rest); rest -= f32(pc); }
                    bits = pc;
                } else bits = bf16(x);
                w |= (unsigned)bits << (16 * half);
            }
            h[(side * 1024 + v) * 4 + e] = w;
        }
    u32x4 *in; float *out; unsigned long long *ticks;
    hipMalloc(&in, h.size() * 4); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&ticks, 2 * 1024 * 8);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>(in, out, ticks, "zero operands, registers");
    run<1>(in, out, ticks, "random bf16 operands, registers");
    run<2>(in, out, ticks, "piece operands (hi/mid/lo of random floats), registers");
    run<3>(in, out, ticks, "pieces + one ds_read_b128 per MFMA");
    run<4>(in, out, ticks, "pieces + ds_read_b128 + two v_fma_f32 per MFMA");
    run<5>(in, out, ticks, "as 4 + 8 idle issue cycles per MFMA and wave");
    run<6>(in, out, ticks, "as 4 + 24 idle issue cycles per MFMA and wave");
    run<7>(in, out, ticks, "as 4 + 48 idle issue cycles per MFMA and wave");
    run<8>(in, out, ticks, "as 4 + 80 idle issue cycles per MFMA and wave");
    run<9>(in, out, ticks, "as 2, products ordered so that A repeats (hh hm hl mh mm lh)");
    run<10>(in, out, ticks, "as 2, products ordered so that A and B both change every time");
    run<2>(in, out, ticks, "mode 2 again");
    return 0;
}
