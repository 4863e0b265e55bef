// qfa_grads_x.h -- pass 2 (gradients) for N_h <= 16 with EVERY contraction on the bf16 XDL pipe (gfx950).
//
// Why a second form of pass 2 (the f32-MFMA k_grads of qfa_step_kernels.h stays for N_h > 16):
//   * stage 1 ([f^T y | f^T C^-1 f], K = 16 + 136) ran as 38 v_mfma_f32_16x16x4_f32 per 16x16 outputs -- 1 216 cycles on
//     the SIMD's float32 datapath, serialised with the VALU work of stage 2 (tools/ubench/mfma_valu.hip);
//   * stage 3 (F-gradient contraction) ran as 102 K = 16 bf16 MFMAs: the 16x16x16 form keeps the XDL pipe at half rate.
// Here both are six-term split-bf16 products (qfa_common.h) at the pipe's full rate:
//   stage 1  v_mfma_f32_16x16x32_bf16: A = [y | Cinv'] of 16 spectra (static, split once per work item, 72 VGPRs),
//            B = the tile image [F^T | pair products] (static per step, split once by k_prep_pgx), 36 MFMAs / 16 px;
//   stage 3  v_mfma_f32_32x32x16_bf16: per PAIR of spectra G = F_tile (32 px x 16 a) x [Z_s | Z_s'] (16 a x 32), both
//            static; beta is applied to G on the VALU; the gamma term is one more product with K = spectrum.
//            48 + 6 MFMAs of 32 cycles per 32 px (was 204 x 16).
// The static operands of a group of 16 spectra are 162 VGPRs' worth of bf16 pieces -- more than a wave that also does
// the per-pixel arithmetic can hold at two waves per SIMD.  So the work of a group is split between the two waves
// that share a SIMD (512-thread workgroup, waves w and w + 4):
//   wave w     (role A)  stage 1 + stage 2 of tile c     : holds [y | Cinv'] pieces, streams the spectra, VALU-heavy
//   wave w + 4 (role B)  stage 3 of tile c - 1           : holds the Z / p pieces, XDL-heavy
// and beta / gamma travel A -> B through LDS, one barrier per tile.  The two instruction streams overlap on the SIMD
// by themselves (matrix pipe beside VALU), which the hand-woven single-wave form needed sched_barrier fences for.
//
// Tile = 32 pixels.  Pixel index inside a tile: role A's lane (lo = lane & 15, g = lane >> 4) owns pixels 2 lo + h
// (h = 0, 1: one 8-byte load per array and spectrum) of the spectra 4 g + r; half h of the stage-1 image holds the
// pixels 2 lo + h in column lo.
#pragma once
#include "qfa_common.h"
#include "qfa_xdl_kernels.h"      // glds16a / dma_wait / lds_addr: the untracked LDS-DMA and counted-vmcnt helpers

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Memory protocol of the tile loop (the one k_moments_x uses, qfa_xdl_kernels.h): the image DMA and the spectra
// prefetch are asm statements, invisible to hipcc's s_waitcnt bookkeeping -- with a tracked LDS-DMA in flight hipcc
// waits vmcnt(0) at the next use of ANY load result and again at __syncthreads(), which exposed the whole memory
// latency twice per tile (measured with in-kernel stamps: 40 % of role A's step).  Per step a wave issues, in this
// order: [role B: the flush stores/atomics of older tiles] -> the DMA pieces of tile c + 1 -> its arithmetic ->
// [role A: the 16 spectra loads of tile c + 2 into the registers it has just consumed] -> s_waitcnt vmcnt(N) with N
// = the loads issued after the DMA (vmcnt retires in issue order) -> s_waitcnt lgkmcnt(0) -> raw s_barrier.
__device__ __forceinline__ void aload8f(f32x2 &dst, const void *sbase, unsigned voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void aload2b(unsigned &dst, const void *sbase, unsigned voff) {
    asm volatile("global_load_ushort %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// a pointer the compiler can see is wave-uniform (an "s" asm operand needs that; values derived from blockIdx through
// divisions are not always proven uniform)
template <typename T>
__device__ __forceinline__ const T *uniform_ptr(const T *p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                   hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return reinterpret_cast<const T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void step_barrier() {        // LDS writes of this step done, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct GX {                                              // KP = 16
    static constexpr int KP = 16, KK2 = 136;
    static constexpr int NKS = 6;                        // K-steps of stage 1: [y, 0 | 5 x 32 pair products (136 used)]
    static constexpr int S1_HALF = NKS * 3 * 1024;       // bytes of one 16-pixel half of the stage-1 image
    static constexpr int OFF_PO = 2 * S1_HALF;           // float32 Psi[32], omega[32] (natural pixel order)
    static constexpr int IMG_B = OFF_PO + 1024;          // ring entry: stage-1 image + Psi/omega (37 KiB)
    static constexpr int OFF_FP = IMG_B;                 // F as bf16 pieces, A operand of stage 3: [piece][lane][8 a]
    static constexpr int TILE_B = OFF_FP + 3 * 1024;     // 40 KiB per 32-pixel tile in global memory
    static constexpr int NCHUNK = TILE_B / 1024;
    static constexpr int GROW = 20;                      // floats per row of the transposed gamma slot (bank spread)
    // LDS (bytes)
    static constexpr int L_IMG = 0;                                  // [2][IMG_B]
    static constexpr int L_FP = L_IMG + 2 * IMG_B;                   // [3][3 KiB]
    static constexpr int L_BETA = L_FP + 3 * 3072;                   // [2][4][16 s][32 px] float
    static constexpr int L_GAM = L_BETA + 2 * 4 * 2048;              // [2][4][32 rows][GROW] float
    static constexpr int L_PART = L_GAM + 2 * 4 * 32 * GROW * 4;     // [2][4][32 px][16 b] float
    static constexpr int L_PSUM = L_PART + 2 * 4 * 2048;             // [2][4][4 sums][2 h][64 lanes] float
    static constexpr int L_SCAL = L_PSUM + 2 * 4 * 2048;             // [4 waves][3 sums][64 lanes] double (role A)
    static constexpr int L_TOTAL = L_SCAL + 4 * 3 * 64 * 8;
};
static_assert(GX::L_TOTAL <= 160 * 1024, "k_grads_x LDS");

__device__ __forceinline__ f32x16 xdl32(const u32x4 &a, const u32x4 &b, f32x16 c) {     // 32x32x16
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                   0);
}
__device__ __forceinline__ f32x16 xdl32_6(const u32x4 &ah, const u32x4 &am, const u32x4 &al, const u32x4 &bh,
                                          const u32x4 &bm, const u32x4 &bl, f32x16 c) {
    c = xdl32(al, bh, c);
    c = xdl32(ah, bl, c);
    c = xdl32(am, bm, c);
    c = xdl32(am, bh, c);
    c = xdl32(ah, bm, c);
    return xdl32(ah, bh, c);
}
// eight float32 values -> three u32x4 of packed bf16 pieces
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned a, b, c;
        split2(x[2 * q], x[2 * q + 1], a, b, c);
        h[q] = a; m[q] = b; l[q] = c;
    }
}

// ------------------------------------------------------------------------------------------------
// k_prep_pgx : F, Psi, omega -> the pass-2 image, one block per 32-pixel tile.
//   stage-1 part  [half h][K-step ks][piece][lane (g, lo)][8 k]  bf16: B[k = 32 ks + 8 g + j][px = 2 lo + h]
//                 ks = 0: k < 16 -> F[px][k]; ks >= 1: pair q = 32 (ks - 1) + 8 g + j -> F[px][a_q] F[px][b_q]
//   Psi / omega   float32 [32] each
//   stage-3 part  [piece][lane (r, h2)][8 a] bf16: A[px = r][a = 8 h2 + j]
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_prep_pgx(const float *__restrict__ F, const float *__restrict__ Psi,
                                                         const float *__restrict__ omega, int Npix, int Nb, int Nh,
                                                         unsigned char *__restrict__ PGX) {
    unsigned char *tile = PGX + (size_t)blockIdx.x * GX::TILE_B;
    const int p0 = 32 * blockIdx.x;
    __shared__ float f[32][17];
    for (int i = threadIdx.x; i < 32 * 16; i += 256) {
        const int px = i >> 4, a = i & 15;
        f[px][a] = (p0 + px < Npix && a < Nh) ? F[(size_t)(p0 + px) * Nh + a] : 0.f;
    }
    __syncthreads();
    // stage-1 part: 2 halves x 6 K-steps x 64 lanes, 8 values each
    for (int i = threadIdx.x; i < 2 * GX::NKS * 64; i += 256) {
        const int lane = i & 63, ks = (i >> 6) % GX::NKS, h = i / (64 * GX::NKS);
        const int lo = lane & 15, g = lane >> 4, px = 2 * lo + h;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * g + j;
            float x = 0.f;
            if (ks == 0) {
                if (kk < 16) x = f[px][kk];
            } else {
                const int q = 32 * (ks - 1) + kk;
                if (q < GX::KK2) {
                    int a = 0;
                    while (a + 1 < 16 && pair_index(a + 1, a + 1, 16) <= q) ++a;
                    const int b = a + (q - pair_index(a, a, 16));
                    x = f[px][a] * f[px][b];
                }
            }
            v[j] = x;
        }
        u32x4 ph, pm, pl;
        split8(v, ph, pm, pl);
        unsigned char *dst = tile + h * GX::S1_HALF + ks * 3072 + lane * 16;
        *reinterpret_cast<u32x4 *>(dst) = ph;
        *reinterpret_cast<u32x4 *>(dst + 1024) = pm;
        *reinterpret_cast<u32x4 *>(dst + 2048) = pl;
    }
    // Psi, omega (+ zero padding of the KiB)
    float *po = reinterpret_cast<float *>(tile + GX::OFF_PO);
    for (int i = threadIdx.x; i < 256; i += 256) {
        const int px = p0 + (i & 31);
        float v = 0.f;
        if (i < 32) v = px < Npix ? Psi[px] : 0.f;
        else if (i < 64) v = px < Nb ? omega[px] : 0.f;
        po[i] = v;
    }
    // stage-3 part
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, r = lane & 31, h2 = lane >> 5;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[r][8 * h2 + j];
        u32x4 ph, pm, pl;
        split8(v, ph, pm, pl);
        unsigned char *dst = tile + GX::OFF_FP + lane * 16;
        *reinterpret_cast<u32x4 *>(dst) = ph;
        *reinterpret_cast<u32x4 *>(dst + 1024) = pm;
        *reinterpret_cast<u32x4 *>(dst + 2048) = pl;
    }
}

struct __attribute__((packed, aligned(4))) f2u { float v[2]; };       // 4-byte aligned 8-byte load
struct __attribute__((packed, aligned(1))) u2u { unsigned char v[2]; };

struct SpecA {                       // role A: one lane's 4 spectra x 2 pixels of a tile
    f32x2 d[4], sg[4], z[4];
    unsigned m[4];                   // 2 mask bytes
};
// No "landing" pins on these registers: the loads of a step are its LAST statements before the counted wait and the
// barrier, and every use sits in a later iteration of the tile loop, so no use can be scheduled between a load and
// its wait.  (Pins -- asm volatile("" : "+v"(reg)) at the first use, as k_moments_x has them -- made the allocator
// spill 190 registers here.)  tools/audit_asm_loads.py checks the compiled code for that property: no instruction
// may read the destination of an asm load before the next s_waitcnt vmcnt.

#ifdef QFA_GX_STAMPS      // diagnostic build only: where a tile step spends its cycles (never shipped)
__device__ unsigned long long qfa_gx_stamps[32];
#define GX_STAMP(var)                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);
#else
#define GX_STAMP(var)
#endif

// ------------------------------------------------------------------------------------------------
// k_grads_x.  One work item = (block of 64 spectra, range of 32-pixel tiles) as in the other passes (WorkPlan).
// ------------------------------------------------------------------------------------------------
template <bool HASA>
__global__ __launch_bounds__(512, 2) void k_grads_x(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau, int B, int Npix, int Nb,
                                                    int Nh, int ntiles, WorkPlan wp,
                                                    const unsigned char *__restrict__ PGX,
                                                    const float *__restrict__ SOL, float *__restrict__ accum,
                                                    float *__restrict__ slab, double *__restrict__ slabS) {
    // slab != NULL: deterministic mode (see k_grads / k_reduce_slab in qfa_step_kernels.h)
    using C = Cfg<16>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[GX::L_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv8 = wave_uniform(tid >> 6);               // 0..7
    const bool roleA = wv8 < 4;
    const int w = wv8 & 3;                                 // group of 16 spectra inside the block
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int s0 = (blk * 4 + w) * 16;
    const bool active = s0 < B;                            // wave-uniform
    const int n = t1 - t0;
    const int nbt = (Nb + 31) >> 5;                        // tiles that contain blue pixels
    const DevConsts k = load_consts(p, tau);

    const bool det = slab != nullptr;
    float *accF = det ? slab + (size_t)blk * ((size_t)Npix * Nh + 3 * (size_t)Npix + Nb) : accum;
    float *accA = accF + (size_t)Npix * Nh;                // sumA | gPsi | gOmega | cnt (contiguous)
    float *accS = accum + (size_t)Npix * Nh + 3 * (size_t)Npix + Nb;
    auto add_to = [&](float *q, float v) {
        if (det) *q = v;                                   // every (block, tile) element is written exactly once
        else atomicAdd(q, v);
    };

    // zero the slots that inactive groups never write
    for (int i = tid; i < (GX::L_TOTAL - GX::L_BETA) / 4; i += 512) reinterpret_cast<float *>(lds + GX::L_BETA)[i] = 0.f;

    // de-phase the tile order between workgroups (concurrent flushes then hit different rows; the workgroups running
    // together still share a window of the image in L2)
#ifdef QFA_GX_NOROT
    const int rot = 0;
#else
    const int rot = n > 0 ? (int)(((unsigned)blk * 2654435761u) % (unsigned)min(n, 32)) : 0;
#endif
    auto tile_of = [&](int c) {
        int x = c + rot;
        if (x >= n) x -= n;
        return t0 + x;
    };
    // LDS-DMA of image tile c by the four role-B waves: wave w moves the 1-KiB pieces w, w + 4, ... (40 pieces, 10 per
    // wave).  Role A issues none: its queue holds its spectra loads only, so its counted wait sits in the MIDDLE of its
    // step (behind stage 1) instead of in front of the barrier.
    auto get_tile = [&](int c) {
        const unsigned char *src = PGX + (size_t)tile_of(c) * GX::TILE_B + lane * 16;
        unsigned char *img = lds + GX::L_IMG + (c & 1) * GX::IMG_B;
        unsigned char *fp = lds + GX::L_FP + (c % 3) * 3072;
        const unsigned char *sbase = uniform_ptr(PGX + (size_t)tile_of(c) * GX::TILE_B);
        (void)src;
#pragma unroll
        for (int i = 0; i < GX::NCHUNK / 4; ++i) {
            const int ch = w + 4 * i;
            unsigned char *dst = ch < GX::IMG_B / 1024 ? img + ch * 1024 : fp + (ch - GX::IMG_B / 1024) * 1024;
            glds16a(sbase + ch * 1024, (unsigned)lane * 16u, wave_uniform(lds_addr(dst)));
        }
    };
    // tile tg leaves the workgroup: thread (px = tid >> 4, b = tid & 15) sums the four groups' partials (fixed order)
    // and adds them to the packed buffer: a wave's 64 lanes cover 4 pixel rows = 256 contiguous bytes at N_h = 16
    // (role B's 256 threads: two outputs each)
    auto flush_F = [&](int tg, int buf) {
        const float *pp = reinterpret_cast<const float *>(lds + GX::L_PART + buf * 4 * 2048);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int o = (tid & 255) + 256 * k2;
            const float v = (pp[o] + pp[512 + o]) + (pp[1024 + o] + pp[1536 + o]);
            const int px = 32 * tg + (o >> 4), b = o & 15;
            if ((b < Nh) & (px < Npix)) add_to(accF + (size_t)px * Nh + b, v);
        }
    };
    // per-pixel sums [sumA | gPsi | gOmega | cnt] of tile tg: thread (which = tid >> 5, pxl = tid & 31), tid < 128
    auto flush_P = [&](int tg, int buf) {
        if ((tid & 255) < 128) {
            const int which = (tid & 255) >> 5, pxl = tid & 31, lo = pxl >> 1, h = pxl & 1;
            const float *q = reinterpret_cast<const float *>(lds + GX::L_PSUM + buf * 4 * 2048) + which * 128 + h * 64 + lo;
            float v = 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) v += (q[g4 * 512] + q[g4 * 512 + 16]) + (q[g4 * 512 + 32] + q[g4 * 512 + 48]);
            const int px = 32 * tg + pxl;
            const int off = which * Npix - (which == 3 ? Npix - Nb : 0) + px;
            const bool ok = (px < Npix) & ((which != 2) | (px < Nb));
            if (ok) add_to(accA + off, v);
        }
    };

#ifndef QFA_GX_ROLE
#define QFA_GX_ROLE 0      // register-pressure experiments: 1 = role A only, 2 = role B only
#endif
    if (roleA && QFA_GX_ROLE != 2) {
#ifdef QFA_GX_PRIO
        __builtin_amdgcn_s_setprio(QFA_GX_PRIO);       // static priority of the VALU-heavy role on its SIMD
#endif
        // ================================================================ role A: stage 1 + stage 2
        const int lo = lane & 15, g = lane >> 4;
        // A operand of stage 1: spectrum s0 + lo, k = 32 ks + 8 g + j
        u32x4 S1h[GX::NKS], S1m[GX::NKS], S1l[GX::NKS];
        {
            const bool v = active && (s0 + lo) < B;
            const float *sol = SOL + (size_t)(v ? s0 + lo : 0) * C::NSOL;
#pragma unroll
            for (int ks = 0; ks < GX::NKS; ++ks) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = 8 * g + j;
                    float val = 0.f;
                    if (ks == 0) {
                        if (v && kk < 16) val = sol[kk];
                    } else {
                        const int q = 32 * (ks - 1) + kk;
                        if (v && q < GX::KK2) val = sol[C::SOL_CI + q];
                    }
                    x[j] = val;
                }
                split8(x, S1h[ks], S1m[ks], S1l[ks]);
            }
        }
        bool sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = active && (s0 + 4 * g + r) < B;
        // row offsets of the lane's four spectra are recomputed where they are used (a handful of VALU instructions
        // per tile) instead of living in eight registers for the whole loop: this role is at the register limit
        const int last_row = active ? min(15, B - 1 - s0) : 0;          // wave-uniform
        const int g4 = 4 * g;
        auto row_of = [&](int r) {
            int q = g4;
            asm volatile("" : "+v"(q));                                // keep the product out of the loop-invariant set
            return (unsigned)min(q + r, last_row);
        };
        auto offN_of = [&](int r) { return row_of(r) * (unsigned)Npix; };
        auto offB_of = [&](int r) { return row_of(r) * (unsigned)Nb; };
        const float *dbase = uniform_ptr(bt.delta + (size_t)(active ? s0 : 0) * Npix);
        const float *ebase = uniform_ptr(bt.error + (size_t)(active ? s0 : 0) * Npix);
        const uint8_t *mbase = uniform_ptr(bt.mask + (size_t)(active ? s0 : 0) * Npix);
        const float *zbase = uniform_ptr(bt.zabs + (size_t)(active ? s0 : 0) * Nb);
        const float *abase = bt.A_blue ? bt.A_blue + (size_t)(active ? s0 : 0) * Nb : nullptr;
        // scalar-gradient sums: float32 inside a tile, float64 across tiles -- the float64 running sums live in LDS
        // (three doubles per lane), not in six registers
        double *scal = reinterpret_cast<double *>(lds + GX::L_SCAL) + (size_t)w * 3 * 64 + lane;
        scal[0] = 0.0; scal[64] = 0.0; scal[128] = 0.0;

        // Spectra of one tile: 16 loads per lane (4 spectra x {delta, sigma, zabs: 8 bytes; mask: 2 bytes}), wave-uniform
        // base in SGPRs + a 32-bit byte offset per lane.  Fast path: asm loads (see the protocol note at the top of
        // the file); returns true.  Ragged end of the pixel axis / of the blue side: ordinary loads, retired on the spot.
        // zabs is loaded for EVERY tile (red tiles re-read the last blue pixels: cache hits, values unused) so that the
        // load count per step is fixed.
        auto load_spec = [&](int tg, SpecA &rg) -> bool {
#ifdef QFA_GX_LOADT0      // timing only: every step re-reads the first tile (cache hits)
            tg = t0;
#endif
            const int pb = 32 * tg + 2 * lo;
            const int tz = min(tg, nbt - 1), pz = 32 * tz + 2 * lo;
            const bool fast = (32 * tg + 31 < Npix) && (Nb == 0 || 32 * tz + 31 < Nb);       // wave-uniform
            if (fast) {
                // (no blue side: zabs is NULL, any valid address keeps the load count; branch-free on purpose)
                const float *zb = Nb > 0 ? zbase : dbase;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned o = offN_of(r) + (unsigned)pb;
                    const unsigned oz = Nb > 0 ? offB_of(r) + (unsigned)pz : o;
                    aload8f(rg.d[r], dbase, 4u * o);
                    aload8f(rg.sg[r], ebase, 4u * o);
                    aload2b(rg.m[r], mbase, o);
                    aload8f(rg.z[r], zb, 4u * oz);
                }
                return true;
            }
            // ragged end: ordinary loads into temporaries, retired HERE (the empty asm reads them), then handed over --
            // a tracked load still pending at the join would make hipcc wait vmcnt(0) in front of the fast path's next
            // (untracked) loads into the same registers
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 td, ts, tzv;
                unsigned mm = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned o = offN_of(r) + (unsigned)min(pb + h, Npix - 1);
                    td[h] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(dbase) + 4u * o);
                    ts[h] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ebase) + 4u * o);
                    mm |= (pb + h < Npix && mbase[o] != 0) ? (1u << (8 * h)) : 0u;
                    tzv[h] = Nb > 0 ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(zbase) +
                                                                       4u * (offB_of(r) + (unsigned)min(pz + h, Nb - 1)))
                                    : 0.f;
                }
                asm volatile("" : "+v"(td), "+v"(ts), "+v"(tzv), "+v"(mm));
                rg.d[r] = td; rg.sg[r] = ts; rg.z[r] = tzv; rg.m[r] = mm;
            }
            return false;
        };

        // stage 1 + stage 2 of one tile
        auto tileA = [&](auto blue_tag, int tg, const SpecA &cur, int buf, bool next_counted) {
            constexpr bool BLUE = decltype(blue_tag)::value;
            const unsigned char *img = lds + GX::L_IMG + buf * GX::IMG_B;
            float *bslot = reinterpret_cast<float *>(lds + GX::L_BETA + (buf * 4 + w) * 2048);
            float *gslot = reinterpret_cast<float *>(lds + GX::L_GAM + (buf * 4 + w) * 32 * GX::GROW * 4);
            float *psum = reinterpret_cast<float *>(lds + GX::L_PSUM + (buf * 4 + w) * 2048);
            float t_tau0 = 0.f, t_c0 = 0.f, t_beta = 0.f;
            const float *po = reinterpret_cast<const float *>(img + GX::OFF_PO);
            // stage 1 of BOTH 16-pixel halves first (72 MFMAs; needs the image and the static operands only), THEN the wait
            // for this tile's spectra: their loads were issued at the end of step c - 2 and so get a step and a third
            f32x4 afy2[2], aq2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char *bp = img + h * GX::S1_HALF + lane * 16;
                f32x4 afy = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f};
                // B pieces of K-step ks + 1 are read while the six MFMAs of K-step ks run; the fences keep the compiler
                // from reading further ahead (every K-step in flight costs 12 registers)
                u32x4 bq[2][3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) bq[0][pc] = *reinterpret_cast<const u32x4 *>(bp + pc * 1024);
#pragma unroll
                for (int ks = 0; ks < GX::NKS; ++ks) {
                    if (ks + 1 < GX::NKS) {
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
                            bq[(ks + 1) & 1][pc] = *reinterpret_cast<const u32x4 *>(bp + (ks + 1) * 3072 + pc * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const u32x4 &bh = bq[ks & 1][0], &bm = bq[ks & 1][1], &bl = bq[ks & 1][2];
                    if (ks == 0) afy = xdl6(S1h[ks], S1m[ks], S1l[ks], bh, bm, bl, afy);
                    else aq = xdl6(S1h[ks], S1m[ks], S1l[ks], bh, bm, bl, aq);
                    __builtin_amdgcn_sched_barrier(0);
                }
                afy2[h] = afy;
                aq2[h] = aq;
            }
            // all but the loads of the NEXT tile (16, when they were issued by the asm path) have landed after this
            if (next_counted) dma_wait<16>();
            else dma_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 afy = afy2[h], aq = aq2[h];
                const float Psi = po[2 * lo + h], om = po[32 + 2 * lo + h];
                const int px = 32 * tg + 2 * lo + h;
                const bool inb = px < Npix;
                const bool blue = px < Nb;
                float gamR[4];
                float gPsi = 0.f, gOm = 0.f, sA = 0.f, cnt = 0.f;
                float betaR[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool wv_ = inb & sv[r] & (((cur.m[r] >> (8 * h)) & 0xffu) != 0);
                    float dd = wv_ ? cur.d[r][h] : 0.f;
                    const float sg = cur.sg[r][h];
                    if (BLUE) {
                        const float l2 = fast_log2(1.0f + cur.z[r][h]);
                        const float pw = fast_exp2(k.beta * l2);
                        const float tauv = k.t_amp * fast_exp2(k.t_expo * (l2 + k.t_lscale)) + k.t_off;   // QFA/utils.py:105-141
                        float Ab = fast_exp2(-tauv * QFA_LOG2E);                                          // QFA/model.py:125
                        if (HASA) Ab = abase[offB_of(r) + (unsigned)min(px, Nb - 1)];                        // custom tau callable
                        const float re = 1.0f - k.c0 - fast_exp2(-k.tau0 * pw * QFA_LOG2E);               // QFA/utils.py:91
                        const float Av = blue ? Ab : 1.f;
                        const float zd = blue ? re * re : 0.f;
                        const float A2 = Av * Av;
                        const float D = A2 * Psi + om * zd + sg * sg;
                        const float wD = wv_ ? fast_rcp(D) : 0.f;
                        const float wDA = wD * Av;
                        const float uu = wD * (dd - Av * afy[r]);                   // (Sigma^-1 delta)_i
                        const float dS = wD - wDA * wDA * aq[r];                    // diag(Sigma^-1)_i
                        const float dG = 0.5f * (dS - uu * uu);                        // QFA/model.py:136,138
                        gPsi += A2 * dG;                                               // :139
                        gOm += dG * zd;                                                // :140
                        const float root = 1.0f - k.tau0 * pw - k.c0;                  // :141
                        const float e = dG * (om * zd) * zd * 2.0f * root;
                        t_tau0 -= e * pw;                                              // :142
                        t_beta -= e * (k.tau0 * pw * (l2 * QFA_LN2));                  // :143
                        t_c0 -= e;                                                     // :144
                        cnt += wv_ ? 1.f : 0.f;
                        betaR[r] = wDA * Av;
                        sA += betaR[r] * Av;
                        gamR[r] = Av * uu;
                    } else {                                                           // red side: A = 1, zd = 0
                        const float D = Psi + sg * sg;
                        const float wD = wv_ ? fast_rcp(D) : 0.f;
                        const float uu = wD * (dd - afy[r]);
                        const float dS = wD - wD * wD * aq[r];
                        gPsi += 0.5f * (dS - uu * uu);
                        cnt += wv_ ? 1.f : 0.f;
                        betaR[r] = wD;
                        sA += wD;
                        gamR[r] = uu;
                    }
#ifndef QFA_GX_NOFENCE
                    if (r & 1) __builtin_amdgcn_sched_barrier(0);      // two elements at a time: bounds the live temporaries
#endif
                }
                // beta[s = 4g + r][pxl = 2 lo + h]
#pragma unroll
                for (int r = 0; r < 4; ++r) bslot[(4 * g + r) * 32 + 2 * lo + h] = betaR[r];
                // gamma transposed: row rho = 16 h + lo, columns s = 4g .. 4g + 3 (one 16-byte store)
                *reinterpret_cast<float4 *>(gslot + (16 * h + lo) * GX::GROW + 4 * g) =
                    float4{gamR[0], gamR[1], gamR[2], gamR[3]};
                psum[0 * 128 + h * 64 + lane] = sA;
                psum[1 * 128 + h * 64 + lane] = gPsi;
                psum[2 * 128 + h * 64 + lane] = gOm;
                psum[3 * 128 + h * 64 + lane] = cnt;
            }
            if (BLUE) {
                scal[0] += (double)t_tau0;
                scal[64] += (double)t_c0;
                scal[128] += (double)t_beta;
            }
        };

        SpecA ra, rb;                        // ra: tiles c = 0, 2, 4 ...; rb: the odd ones; each is reloaded two tiles ahead
        bool cnt_a = false, cnt_b = false;   // were the loads now in flight into ra / rb issued by the (counted) asm path
#ifdef QFA_GX_STAMPS
        unsigned long long st_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
        if (n > 0 && active) {
            cnt_a = load_spec(tile_of(0), ra);
            if (n > 1) cnt_b = load_spec(tile_of(1), rb);
        }
        step_barrier();                      // (role B's wait in front of this barrier covers the image of tile 0)
        // one step: stage 1, wait for the spectra of this tile, stage 2, refill the registers with tile c + 2
        auto stepA = [&](int c, SpecA &cur, bool &cnt_cur, bool cnt_other) {
#ifdef QFA_GX_STAMPS
            unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
#endif
            GX_STAMP(q0)
            GX_STAMP(q1)
            if (c < n && active) {
                const int tg = tile_of(c);
                const bool nc = (c + 1 < n) && cnt_other;
                if (tg < nbt) tileA(std::true_type{}, tg, cur, c & 1, nc);
                else tileA(std::false_type{}, tg, cur, c & 1, nc);
            }
            GX_STAMP(q2)
            cnt_cur = false;
#ifdef QFA_GX_NOLOAD      // timing only
            if (c + 2 < n && active && c < 2) cnt_cur = load_spec(tile_of(c + 2), cur);
#else
            if (c + 2 < n && active) cnt_cur = load_spec(tile_of(c + 2), cur);
#endif
            GX_STAMP(q3)
            step_barrier();
#ifdef QFA_GX_STAMPS
            GX_STAMP(q4)
            if (c >= 2 && c < n) {
                const int o = tile_of(c) < nbt ? 0 : 8;
                st_t[o + 0] += q1 - q0; st_t[o + 1] += q2 - q1; st_t[o + 2] += q3 - q2; st_t[o + 3] += q4 - q3; st_t[o + 4] += 1;
            }
#endif
        };
        for (int c = 0; c < n + 2; c += 2) {
            stepA(c, ra, cnt_a, cnt_b);
            if (c + 1 < n + 2) stepA(c + 1, rb, cnt_b, cnt_a);
        }
        dma_wait<0>();
#ifdef QFA_GX_STAMPS
        if (blk == 300 && w == 0 && lane == 0 && seg == 0)
            for (int i = 0; i < 16; ++i) qfa_gx_stamps[i] = st_t[i];
#endif
        double s_tau0 = scal[0], s_c0 = scal[64], s_beta = scal[128];
        if (active) {
            for (int o = 32; o >= 1; o >>= 1) {
                s_tau0 += __shfl_xor(s_tau0, o);
                s_c0 += __shfl_xor(s_c0, o);
                s_beta += __shfl_xor(s_beta, o);
            }
            if (lane == 0) {
                if (det) {
                    double *q = slabS + ((size_t)blockIdx.x * 4 + w) * 3;
                    q[0] = s_tau0; q[1] = s_c0; q[2] = s_beta;
                } else {
                    atomicAdd(accS + 0, (float)s_tau0);
                    atomicAdd(accS + 1, (float)s_c0);
                    atomicAdd(accS + 2, (float)s_beta);
                }
            }
        } else if (det && lane == 0) {
            double *q = slabS + ((size_t)blockIdx.x * 4 + w) * 3;
            q[0] = 0.0; q[1] = 0.0; q[2] = 0.0;
        }
    } else if (QFA_GX_ROLE != 1) {
        // ================================================================ role B: stage 3
        const int col = lane & 31, h2 = lane >> 5, b = lane & 15, sp = (lane >> 4) & 1;
        // B operands: Z of the pair (2p, 2p + 1): B[k = a = 8 h2 + j][col = (sp, b)] = Z_{2p + sp}[a][b]
        u32x4 Zh[8], Zm[8], Zl[8], Ph, Pm, Pl;
        {
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                const int s = s0 + 2 * pr + sp;
                const bool v = active && s < B && b < Nh;
                const float *sol = SOL + (size_t)(v ? s : 0) * C::NSOL + C::SOL_Z + b;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = v ? sol[(8 * h2 + j) * 16] : 0.f;
                split8(x, Zh[pr], Zm[pr], Zl[pr]);
            }
            // gamma term: B[k = s = 8 h2 + j][col] = p_s[b] for col < 16, 0 otherwise
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = s0 + 8 * h2 + j;
                const bool v = active && s < B && col < 16 && b < Nh;
                x[j] = v ? SOL[(size_t)s * C::NSOL + C::SOL_P + b] : 0.f;
            }
            split8(x, Ph, Pm, Pl);
        }
#ifndef QFA_GX_PFD
#define QFA_GX_PFD 3          // prefetch distance in tiles
#endif
        // lane (row = lane & 15, arr = lane >> 4) touches the segment of array arr = {delta, sigma, zabs, mask} of
        // spectrum s0 + row that tile c covers; pf_sink is the (never read) destination, live for the whole loop
        unsigned pf_sink = 0;
        const unsigned char *pf_base;
        unsigned pf_stride, pf_esize, pf_len;          // row stride, bytes per pixel, pixels on the axis of this array
        {
            const int arr = lane >> 4, row = active ? min(lane & 15, B - 1 - s0) : 0;
            const size_t s = (size_t)(active ? s0 : 0) + row;
            if (arr == 0) { pf_base = reinterpret_cast<const unsigned char *>(bt.delta + s * Npix); pf_esize = 4; pf_len = Npix; }
            else if (arr == 1) { pf_base = reinterpret_cast<const unsigned char *>(bt.error + s * Npix); pf_esize = 4; pf_len = Npix; }
            else if (arr == 2 && Nb > 0) { pf_base = reinterpret_cast<const unsigned char *>(bt.zabs + s * Nb); pf_esize = 4; pf_len = Nb; }
            else { pf_base = reinterpret_cast<const unsigned char *>(bt.mask + s * Npix); pf_esize = 1; pf_len = Npix; }
            pf_stride = 0;
        }
        auto prefetch = [&](int c) {
            const int tg = n > 0 ? tile_of(max(c, 0)) : 0;
            const unsigned p0 = (unsigned)min(32 * tg, (int)pf_len - 1), p1 = (unsigned)min(32 * tg + 31, (int)pf_len - 1);
            const unsigned char *a0 = pf_base + (size_t)p0 * pf_esize, *a1 = pf_base + (size_t)p1 * pf_esize + (pf_esize - 1);
            asm volatile("global_load_ubyte %0, %1, off" : "+v"(pf_sink) : "v"(a0) : "memory");
            asm volatile("global_load_ubyte %0, %1, off" : "+v"(pf_sink) : "v"(a1) : "memory");
        };
        (void)pf_stride;
        auto tileB = [&](int c) {
            const int buf = c & 1;
            const unsigned char *fp = lds + GX::L_FP + (c % 3) * 3072 + lane * 16;
            const float *bslot = reinterpret_cast<const float *>(lds + GX::L_BETA + (buf * 4 + w) * 2048);
            const float *gslot = reinterpret_cast<const float *>(lds + GX::L_GAM + (buf * 4 + w) * 32 * GX::GROW * 4);
            float *part = reinterpret_cast<float *>(lds + GX::L_PART + (buf * 4 + w) * 2048);
            const u32x4 Fh = *reinterpret_cast<const u32x4 *>(fp), Fm = *reinterpret_cast<const u32x4 *>(fp + 1024),
                        Fl = *reinterpret_cast<const u32x4 *>(fp + 2048);
            // gamma of pixel pxl = col (row rho = 16 (col & 1) + (col >> 1)), spectra 8 h2 .. 8 h2 + 7
            const float *grow = gslot + (16 * (col & 1) + (col >> 1)) * GX::GROW + 8 * h2;
            const float4 g0 = *reinterpret_cast<const float4 *>(grow), g1 = *reinterpret_cast<const float4 *>(grow + 4);
            const float gx[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            u32x4 Gh, Gm, Gl;
            split8(gx, Gh, Gm, Gl);
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;
            f32x16 acc = xdl32_6(Gh, Gm, Gl, Ph, Pm, Pl, zero);
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                const f32x16 G = xdl32_6(Fh, Fm, Fl, Zh[pr], Zm[pr], Zl[pr], zero);
                const float *brow = bslot + (2 * pr + sp) * 32 + 4 * h2;       // pixels 8 q + 4 h2 + (0..3)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bq = *reinterpret_cast<const float4 *>(brow + 8 * q);
                    acc[4 * q + 0] = fmaf(bq.x, G[4 * q + 0], acc[4 * q + 0]);
                    acc[4 * q + 1] = fmaf(bq.y, G[4 * q + 1], acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(bq.z, G[4 * q + 2], acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(bq.w, G[4 * q + 3], acc[4 * q + 3]);
                }
            }
            // sum the two spectra of the pairs (lanes l and l ^ 16), then each half of the lanes stores half the rows:
            // part[pxl][b], pxl = (i & 3) + 8 (i >> 2) + 4 h2
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(acc[i]), 0x401f));   // xor 16
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ii = sp ? 8 + i : i;
                const float v = sp ? acc[8 + i] : acc[i];
                part[((ii & 3) + 8 * (ii >> 2) + 4 * h2) * 16 + b] = v;
            }
        };
#ifdef QFA_GX_STAMPS
        unsigned long long st_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        if (n > 0) get_tile(0);
        dma_wait<0>();
        step_barrier();
        for (int c = 0; c < n + 2; ++c) {
#ifdef QFA_GX_STAMPS
            unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
#endif
            GX_STAMP(q0)
            // the flushes first: their stores / atomics are the oldest requests of the step and have the whole of
            // stage 3 to drain before the vmcnt(0) in front of the barrier
#ifndef QFA_GX_NOFLUSH    // (defined: timing only)
            if (c >= 1 && c <= n) flush_P(tile_of(c - 1), (c - 1) & 1);
            if (c >= 2) flush_F(tile_of(c - 2), c & 1);
#endif
            GX_STAMP(q1)
#ifdef QFA_GX_NODMA       // timing only
            if (c + 1 < n && c < 2) get_tile(c + 1);
#else
            if (c + 1 < n) get_tile(c + 1);
#endif
            GX_STAMP(q2)
            if (c >= 1 && c <= n && active) tileB(c - 1);
#ifndef QFA_GX_NOPF
            // L2 prefetch of the spectra role A will load QFA_GX_PFD tiles from now: two loads per lane (first and last
            // byte of its row segment: every 128-byte line of it), results never read.  They are the youngest requests
            // of the step: the counted wait below leaves them in flight.
            prefetch(c + QFA_GX_PFD < n ? c + QFA_GX_PFD : n - 1);
            dma_wait<2>();                         // everything but the two prefetch loads: retires the DMA of tile c + 1
#else
            dma_wait<0>();
#endif
            GX_STAMP(q3)
            step_barrier();
#ifdef QFA_GX_STAMPS
            GX_STAMP(q4)
            if (c >= 2 && c < n) { st_t[0] += q1 - q0; st_t[1] += q2 - q1; st_t[2] += q3 - q2; st_t[3] += q4 - q3; st_t[4] += 1; }
#endif
        }
        dma_wait<0>();
        asm volatile("" ::"v"(pf_sink));               // the prefetch destination stayed reserved up to here
#ifdef QFA_GX_STAMPS
        if (blk == 300 && w == 0 && lane == 0 && seg == 0)
            for (int i = 0; i < 8; ++i) qfa_gx_stamps[16 + i] = st_t[i];
#endif
    }
}
