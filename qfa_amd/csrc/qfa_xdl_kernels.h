// qfa_xdl_kernels.h -- pass 1 on the bf16 matrix pipe (XDL) of gfx950, at float32 accuracy (the split / MFMA helpers
// live in qfa_common.h; stage 3 of pass 2 uses them too, qfa_step_kernels.h).
//
// Measured on MI355X (tools/ubench/mfma_valu.hip): v_mfma_f32_16x16x4_f32 shares the SIMD's float32 datapath with
// the VALU -- an f32 MFMA and the VALU instructions around it serialise (32 + 4.5 n cycles for one MFMA and n
// fmas) -- while v_mfma_f32_16x16x32_bf16 runs on the XDL pipe beside the VALU, takes 16 cycles and carries 8x
// the K.  So every float32 operand x is split into three bf16 pieces x = h + m + l (round-to-nearest each, the
// residuals are exact in float32) and a product sum_k a_k b_k is issued as the six XDL MFMAs
//      al.bh  ah.bl  am.bm  am.bh  ah.bm  ah.bh       (the dropped terms are <= 2^-24 relative)
// accumulating in float32.  tools/ubench/bf16x3_numerics.hip: the error against float64 is at or below that of
// the f32 MFMA / fmaf chain for K = 32..4096, signed, positive and wide-range operands (the 3-term variant is not:
// 1e-6..1e-5).  The static operands (the parameter image) are split once per step by k_prep_pfx; the per-element
// weights are split in the loop (11 VALU instructions per pair of values).
//
//   k_prep_pfx    F, Psi, omega -> PFX image: per 32-pixel tile [piece h|m|l][16-column tile][8-px group][column][8 px]
//                 bf16 + Psi, omega
//   k_moments_x   pass 1 (C, T, b, b2 + scalar sums) with K = 32 pixels per MFMA
#pragma once
#include "qfa_common.h"

#ifndef QFA_P1_F16
#define QFA_P1_F16 0       // pass 1 with the PAIR columns on two float16 pieces (C, T: three products; b, b2 stay on bf16).  Built, measured and
                           // NOT shipped (round 5, profiles/r5_ab_f16_pass1.txt): pass 1 at c3 1.165 -> 1.005 ms, step 2.95 -> 2.79, but the
                           // moments feed the k x k solve: the normalised F gradient against the float64 oracle goes from 1.6e-5 to 3.1e-5
                           // at 25 000 spectra (6.4e-5 -> 6.9e-5 at 100 000; NLL, gPsi, gOmega unchanged) -- a contribution of 2.6e-5 that
                           // does not average down with the number of spectra and is NOT the image's 22 bits (a third, exact float16 piece of
                           // the image and a fourth product: the same 3.09e-5), and tests/test_stage3_precision.py::
                           // test_f_columns_spanning_three_decades goes from under its 3e-4 bar to 3.5e-4.  The six bf16 products are BETTER
                           // than a float32 MFMA (1.0e-8 against 2.9e-8 per 32-term sum), the three float16 ones equal to it (2.8e-8):
                           // stages 1 and 3 of pass 2 do not notice, the solve does.  Two more launches per step (memset + k_colmax) cost a
                           // small batch 4 % (c2).
#endif
template <int KP>
struct XCfg {
    using C = Cfg<KP>;
    static constexpr int NCOL = C::FW + C::PW;               // f columns first, then the pair columns
    static constexpr int NCT = NCOL / 16;                     // 16-column tiles: 4 / 10 / 35
    // N_h > 16: the image of a 32-pixel tile (105 KiB) does not fit LDS twice -- it is cut into NSW = 2 sub-images of
    // column tiles [0, CT1) and [CT1, NCT), which a tile step sweeps one after the other (ring slot 0 / slot 1)
    static constexpr int NSW = KP > 16 ? 2 : 1;
    static constexpr int CT1 = (NCT + NSW - 1) / NSW;
    static constexpr int nct(int j) { return j == 0 ? CT1 : NCT - CT1; }
    static constexpr int pstr(int j) { return nct(j) * 1024; }   // bytes of one piece of sub-image j: 32 px x bf16 per column
    static constexpr int PSTR = pstr(0);
    // QFA_P1_F16 (N_h <= 16; an option, see the macro): the PAIR columns as two float16 pieces (C and T: three products per column
    // tile instead of six), the F columns (b, b2: their weights wD A delta carry the data, no bound) as three bf16 pieces as before.
    // Planes: [h: NCT KiB][m: NCT KiB][l: the NFT tiles of F only].  A pair column holds f_a f_b / s_px x cs: s_px =
    // f16_weight_scale(Psi of the pixel) -- the C / T weights are multiplied by it and stay below 2^12 whatever the data, the two
    // powers cancel inside the product (K = pixel) -- and cs = the column's power of two over the WHOLE pixel axis (k_colmax; its
    // inverse multiplies the moment when it is stored).
    static constexpr bool F16 = QFA_P1_F16 != 0 && NSW == 1;
    static constexpr int OFF_PSI = F16 ? 2 * PSTR + C::NFT * 1024 : 3 * PSTR;
                                                              // in sub-image 0: float32 Psi[32], omega[32], mu[32] (prediction),
                                                              // ti[32], pwi[32], offp[32] (factored-z form: ZP of qfa_common.h;
                                                              // all 0 for red pixels), blue[32] = 1.0 / 0.0, F16: s_px[32]
    static constexpr int SUB0_B = F16 ? (OFF_PSI + 1024 + 1023) / 1024 * 1024 : (3 * PSTR + 896 + 1023) / 1024 * 1024;   // whole 1-KiB LDS-DMA pieces
    static constexpr int SUB1_B = NSW > 1 ? 3 * pstr(1) : 0;
    static constexpr int sub_off(int j) { return j == 0 ? 0 : SUB0_B; }
    static constexpr int sub_bytes(int j) { return j == 0 ? SUB0_B : SUB1_B; }
    static constexpr int TILE_B = SUB0_B + SUB1_B;            // bytes of a tile in global memory
    static constexpr int SLOT_B = SUB0_B > SUB1_B ? SUB0_B : SUB1_B;       // LDS ring slot
    static constexpr int NCHUNK = SUB0_B / 1024;
};

// LDS-DMA: the wave's 64 lanes move 64 x 16 B from per-lane global addresses to LDS byte address lds_dst + 16 * lane
// (lds_dst wave-uniform).  Written as asm so that hipcc does not count it: with a tracked LDS-DMA in flight hipcc
// waits vmcnt(0) at the next use of any load result, which would serialise the spectra prefetch behind the image
// copy.  An untracked extra entry in the in-order vmcnt queue only makes the compiler's own counted waits more
// conservative; its completion is waited for explicitly (dma_wait<N>: all but the N youngest requests retired).
// Addresses are wave-uniform base (SGPR pair) + per-lane 32-bit byte offset (VGPR): no address VALU in the loop.
// M0 (the LDS destination) is written and not restored: a restore behind the DMA waits until the texture path has
// accepted the request (measured: ~150 cycles per piece, 1 250-2 500 per tile); nothing else in this kernel
// depends on M0.
// QFA_TRACKED_LOADS=1 (test build, `make tracked` -> libqfa_tracked.so): every load, LDS-DMA and wait of the XDL
// kernels in the form hipcc keeps the books for -- builtin LDS-DMA, ordinary loads, vmcnt(0) and a full
// __syncthreads() at every hand-over.  Slow, but free of hand-counted waits: tests/test_tracked_loads.py requires
// its results to be bit-identical to the shipped build's, which is what shows the counted waits to be sufficient.
#ifndef QFA_TRACKED_LOADS
#define QFA_TRACKED_LOADS 0
#endif
__device__ __forceinline__ void glds16a(const void *sbase, unsigned voff, unsigned lds_dst) {
#if QFA_TRACKED_LOADS
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(sbase) + voff),
        (__attribute__((address_space(3))) void *)(size_t)__builtin_amdgcn_readfirstlane((int)lds_dst), 16, 0, 0);
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
}
// The same request without the "memory" clobber, for a piece issued in the middle of a stage: the clobber keeps hipcc from
// moving the stage's own LDS reads across the statement, which exposes a full LDS latency at every piece.  Safe where the
// protocol already separates the piece's destination from everything the stage reads (the destination is a ring slot that
// nobody reads before the next counted wait + barrier, which do carry the clobber).
__device__ __forceinline__ void glds16a_nc(const void *sbase, unsigned voff, unsigned lds_dst) {
#if QFA_TRACKED_LOADS
    glds16a(sbase, voff, lds_dst);
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst));
#endif
}
template <int N>
__device__ __forceinline__ void dma_wait() {
#if QFA_TRACKED_LOADS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)p; }   // LDS aperture: low 32 bits
// workgroup barrier behind explicit waits (the caller has waited for what the hand-over needs)
__device__ __forceinline__ void wg_barrier() {
#if QFA_TRACKED_LOADS
    __syncthreads();
#else
    __builtin_amdgcn_s_barrier();
#endif
}
// 4 bytes per lane into LDS (any source alignment: tools/ubench/glds_align.hip); see glds16a for M0
__device__ __forceinline__ void glds4a(const void *sbase, unsigned voff, unsigned lds_dst) {
#if QFA_TRACKED_LOADS
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(sbase) + voff),
        (__attribute__((address_space(3))) void *)(size_t)__builtin_amdgcn_readfirstlane((int)lds_dst), 4, 0, 0);
#else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
}
// The same requests from a per-lane 64-bit address (the spectra rows of the batch: qfa_common.h, batch_row / lane_ptr)
__device__ __forceinline__ void glds16p(const void *ptr, unsigned lds_dst) {
#if QFA_TRACKED_LOADS
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ptr,
                                     (__attribute__((address_space(3))) void *)(size_t)__builtin_amdgcn_readfirstlane((int)lds_dst), 16, 0, 0);
#else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(ptr), "s"(lds_dst) : "memory");
#endif
}
__device__ __forceinline__ void glds4p(const void *ptr, unsigned lds_dst) {
#if QFA_TRACKED_LOADS
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)ptr,
                                     (__attribute__((address_space(3))) void *)(size_t)__builtin_amdgcn_readfirstlane((int)lds_dst), 4, 0, 0);
#else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(ptr), "s"(lds_dst) : "memory");
#endif
}
// N consecutive 1-KiB pieces (global and LDS both contiguous) behind ONE write of M0: the instruction's immediate offset
// moves the global address and the LDS address together.  (A write of M0 behind an LDS-DMA waits until the texture path has
// accepted that request -- ~150 cycles per piece when every piece sets M0 itself, tools/gt_stamps.sh.)  The offsets are
// biased by -2048 so that five pieces fit the signed 13-bit field.
template <int N>
__device__ __forceinline__ void glds16_run(const void *sbase, unsigned voff, unsigned lds_dst) {
    static_assert(N >= 1 && N <= 5, "pieces per run");
#if QFA_TRACKED_LOADS
#pragma unroll
    for (int i = 0; i < N; ++i) glds16a(reinterpret_cast<const unsigned char *>(sbase) + 1024 * i, voff, lds_dst + 1024 * i);
#else
    const unsigned char *sb = reinterpret_cast<const unsigned char *>(sbase) + 2048;
    const unsigned d = lds_dst + 2048;
    if constexpr (N == 1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048" ::"v"(voff), "s"(sb), "s"(d) : "memory");
    else if constexpr (N == 2)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:-1024" ::"v"(voff), "s"(sb), "s"(d) : "memory");
    else if constexpr (N == 3)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:-1024\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(d) : "memory");
    else if constexpr (N == 4)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:-1024\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(sb), "s"(d) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:-1024\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(voff), "s"(sb), "s"(d) : "memory");
#endif
}
template <int N>      // any number of consecutive pieces: runs of at most five
__device__ __forceinline__ void glds16_runs(const void *sbase, unsigned voff, unsigned lds_dst) {
    if constexpr (N <= 5) glds16_run<N>(sbase, voff, lds_dst);
    else {
        glds16_run<5>(sbase, voff, lds_dst);
        glds16_runs<N - 5>(reinterpret_cast<const unsigned char *>(sbase) + 5120, voff, lds_dst + 5120);
    }
}
// s_waitcnt vmcnt(k) for a wave-uniform run-time k (0..63): all but the k youngest vector-memory requests retired
__device__ __forceinline__ void dma_wait_n(int k) {
#if QFA_TRACKED_LOADS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
#define QFA_W1(K) case K: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
#define QFA_W8(K) QFA_W1(K) QFA_W1(K + 1) QFA_W1(K + 2) QFA_W1(K + 3) QFA_W1(K + 4) QFA_W1(K + 5) QFA_W1(K + 6) QFA_W1(K + 7)
    switch (k) {
        QFA_W8(0) QFA_W8(8) QFA_W8(16) QFA_W8(24) QFA_W8(32) QFA_W8(40) QFA_W8(48) QFA_W8(56)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef QFA_W8
#undef QFA_W1
#endif
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
    wg_barrier();
    asm volatile("" ::: "memory");
}


// ------------------------------------------------------------------------------------------------
// The column scales of the pass-1 image (XCfg::F16).  COLMAX (NCOL unsigned behind the last tile of PFX, zeroed by the host) takes
// max_px |f_a f_b| / s_px of every pair column as float32 bits (non-negative floats order like their bits: atomicMax); the image
// builder and the epilogue of pass 1 derive the column's power of two from it (pfx_col_scale: largest entry in [2^13, 2^14)).
// One block per 32-pixel tile, thread = (pair column, 8-pixel group).
template <int KP>
__host__ __device__ inline unsigned *pfx_colmax(unsigned char *PFX, int ntiles32) {
    return reinterpret_cast<unsigned *>(PFX + (size_t)ntiles32 * XCfg<KP>::TILE_B);
}
__device__ __forceinline__ float pfx_col_scale(unsigned maxbits, float &inv) { return f16_row_scale(__uint_as_float(maxbits), inv); }
template <int KP>
__global__ __launch_bounds__(1024) void k_colmax(const float *__restrict__ F, const float *__restrict__ Psi, int Npix, int Nh,
                                                 unsigned *__restrict__ COLMAX) {
    using C = Cfg<KP>;
    __shared__ float f[32][KP + 1], isw[32];
    const int tid = threadIdx.x, p0 = 32 * blockIdx.x;
    for (int i = tid; i < 32 * KP; i += 1024) {
        const int px = i / KP, a = i % KP;
        f[px][a] = (p0 + px < Npix && a < Nh) ? F[(size_t)(p0 + px) * Nh + a] : 0.f;
    }
    if (tid < 32) isw[tid] = 1.f / f16_weight_scale(p0 + tid < Npix ? Psi[p0 + tid] : 1.f);
    __syncthreads();
    for (int i = tid; i < 4 * C::KK2; i += 1024) {
        const int q = i >> 2, g8 = i & 3;
        int a = 0;
        while (a + 1 < KP && pair_index(a + 1, a + 1, KP) <= q) ++a;
        const int b = a + (q - pair_index(a, a, KP));
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(f[8 * g8 + j][a] * f[8 * g8 + j][b]) * isw[8 * g8 + j]);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        if (g8 == 0 && m > 0.f && m < 3.0e38f) atomicMax(COLMAX + C::FW + q, __float_as_uint(m));
    }
}

template <int KP>
__device__ __forceinline__ void prep_pfx_body(int bid, int ntiles32, const float *__restrict__ F, const float *__restrict__ Psi,
                                              const float *__restrict__ omega, const float *__restrict__ mu,
                                              const ZPSrc &ZP, int Npix, int Nb, int Nh, unsigned char *__restrict__ PFX) {
    using C = Cfg<KP>;
    using X = XCfg<KP>;
    unsigned char *tile = PFX + (size_t)bid * X::TILE_B;
    const unsigned *COLMAX = pfx_colmax<KP>(PFX, ntiles32);
    for (int idx = threadIdx.x; idx < X::NCOL * 16; idx += 256) {         // (column, pixel pair)
        const int c = idx >> 4, q = (idx & 15) * 2;
        float v[2] = {0.f, 0.f};
        int a = 0, b = 0;
        bool pairc = false, okc;
        if (c < C::FW) {
            a = c;
            okc = c < Nh;
        } else {
            const int pidx = c - C::FW;
            pairc = true;
            okc = pidx < C::KK2;
            if (okc) {
                while (a + 1 < KP && pair_index(a + 1, a + 1, KP) <= pidx) ++a;
                b = a + (pidx - pair_index(a, a, KP));
                okc = b < Nh;
            }
        }
        float cs = 1.f, ics_ = 1.f;
        if (X::F16 && pairc) cs = pfx_col_scale(COLMAX[c], ics_);
        (void)ics_;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 32 * bid + q + e;
            if (okc && i < Npix) {
                v[e] = pairc ? F[(size_t)i * Nh + a] * F[(size_t)i * Nh + b] : F[(size_t)i * Nh + a];
                if (X::F16 && pairc) v[e] = (v[e] / f16_weight_scale(Psi[i])) * cs;          // (powers of two: exact)
            }
        }
        // lane-linear for the B-operand read: [16-column tile][8-pixel group][column][8 px]
        const int ct = c >> 4, sw = ct >= X::CT1 ? 1 : 0, ctl = ct - (sw ? X::CT1 : 0), ps = sw ? X::pstr(1) : X::pstr(0);
        unsigned *dst = reinterpret_cast<unsigned *>(tile + (sw ? X::SUB0_B : 0) + ctl * 1024 + (q >> 3) * 256 + (c & 15) * 16 +
                                                     (q & 7) * 2);
        if (X::F16 && pairc) {
            unsigned h, m;
            split2h(v[0], v[1], h, m);
            dst[0] = h;
            dst[ps / 4] = m;
        } else {
            unsigned h, m, l;
            split2(v[0], v[1], h, m, l);
            dst[0] = h;
            dst[ps / 4] = m;
            dst[2 * ps / 4] = l;            // (F16: the l plane holds the NFT tiles of F only -- ct < NFT here)
        }
    }
    float *po = reinterpret_cast<float *>(tile + X::OFF_PSI);
    for (int idx = threadIdx.x; idx < (X::SUB0_B - X::OFF_PSI) / 4; idx += 256) {
        const int i = 32 * bid + (idx & 31);
        float v = 0.f;
        if (idx < 32) v = i < Npix ? Psi[i] : 0.f;
        else if (idx < 64) v = i < Nb ? omega[i] : 0.f;
        else if (idx < 96) v = (mu && i < Npix) ? mu[i] : 0.f;      // mean continuum (prediction: delta = flux - mu A)
        else if (idx < 128) v = (ZP.on() && i < Nb) ? ZP.at(i).x : 0.f;     // factored-z form: ti
        else if (idx < 160) v = (ZP.on() && i < Nb) ? ZP.at(i).y : 0.f;     //                  pwi
        else if (idx < 192) v = (ZP.on() && i < Nb) ? ZP.offp : 0.f;        //                  offset of the exponent of A (0: red pixel)
        else if (idx < 224) v = i < Nb ? 1.f : 0.f;                         // 1 = blue pixel
        else if (idx < 256) v = X::F16 ? f16_weight_scale(i < Npix ? Psi[i] : 1.f) : 0.f;      // s_px of the C / T weights
        po[idx] = v;
    }
}
template <int KP>
__global__ __launch_bounds__(256) void k_prep_pfx(const float *__restrict__ F, const float *__restrict__ Psi,
                                                  const float *__restrict__ omega, const float *__restrict__ mu,
                                                  const float4 *__restrict__ ZP, float offp, int Npix, int Nb, int Nh,
                                                  unsigned char *__restrict__ PFX) {
    prep_pfx_body<KP>(blockIdx.x, gridDim.x, F, Psi, omega, mu, zp_table(ZP, offp), Npix, Nb, Nh, PFX);
}

// ------------------------------------------------------------------------------------------------
// k_moments_x (pass 1).  Lane (sl = lane&15, g = lane>>4) owns spectrum s0+sl at the pixels 32*tile + 8g + j
// (j = 0..7): the A-operand layout A[row = lane&15][k = 8(lane>>4) + j] of v_mfma_f32_16x16x32_bf16, and 32
// contiguous bytes per input array.  B[k][col] = image column `col` at the same 8 pixels: one ds_read_b128 per
// piece (lanes of a wave read 1 KiB contiguously).  The 31-KiB image tile of the next step is moved by LDS-DMA
// (no staging registers) while the current one is consumed; one barrier per tile.
// Red tiles first (A = 1: T and b2 receive what C and b receive), then the blue tiles.
// ------------------------------------------------------------------------------------------------
struct Pieces {         // bf16 pieces (h, m, l) of the four weight vectors of a lane's 8 pixels
    u32x4 w1h, w1m, w1l, w2h, w2m, w2l, w3h, w3m, w3l, w4h, w4m, w4l;
};

struct SpecRegsX {        // one lane's 8 pixels of a tile: delta, sigma, zabs, 8 mask bytes
    f32x4 d0, d1, s0, s1, z0, z1;
    u32x2 m;
};
// Spectra loads as asm: hipcc cannot keep counted vmcnt waits across the loop back-edge (it falls back to
// vmcnt(0) at the first use, which would also wait for the tile requested last), so these loads are invisible to
// its bookkeeping and retired by dma_wait<N>() in front of the tile barrier.  land() makes every later use of the
// registers depend on a point behind that wait (the compiler may otherwise schedule a use right after the load).
template <int OFF>
__device__ __forceinline__ void aload16(f32x4 &dst, const void *sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
__device__ __forceinline__ void aload8(u32x2 &dst, const void *sbase, unsigned voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// the same loads from a per-lane 64-bit address (the spectra rows: qfa_common.h, lane_ptr)
template <int OFF>
__device__ __forceinline__ void aload16p(f32x4 &dst, const void *ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void aload8p(u32x2 &dst, const void *ptr) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void aload4(float &dst, const void *sbase, unsigned voff) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <bool WITHZ>
__device__ __forceinline__ void land(SpecRegsX &r) {
    asm volatile("" : "+v"(r.d0), "+v"(r.d1), "+v"(r.s0), "+v"(r.s1), "+v"(r.m));
    if (WITHZ) asm volatile("" : "+v"(r.z0), "+v"(r.z1));
}

#ifndef QFA_P1_EARLY_DMA
#define QFA_P1_EARLY_DMA 0  // 1: image DMA of tile c + 1 at the START of step c instead of behind the weights (measured: 1.35 against 1.32 ms)
#endif
#ifndef QFA_P1_ABL
#define QFA_P1_ABL 0        // timing-only ablations of k_moments_x (N_h <= 16): 1 no spectra reloads, 2 no MFMAs, 4 no weights, 8 no image DMA
#endif
#ifndef QFA_P1_CT_TERMS
#define QFA_P1_CT_TERMS 6      // experiment: 4 = C and T from the two leading pieces of both operands
#endif
__device__ __forceinline__ f32x4 xdl_ct(const u32x4 &ah, const u32x4 &am, const u32x4 &al, const u32x4 &bh,
                                        const u32x4 &bm, const u32x4 &bl, f32x4 c) {
    if (QFA_P1_CT_TERMS == 6) return xdl6(ah, am, al, bh, bm, bl, c);
    c = xdl(am, bm, c);
    c = xdl(am, bh, c);
    c = xdl(ah, bm, c);
    return xdl(ah, bh, c);
}
// Fresh accumulators per tile (round 5).  v_mfma_f32_16x16x32_bf16 aligns its 32 products with the accumulator and truncates what
// lies ~2 bits below its last place: a BIAS per instruction of ~2^-26 of the accumulator.  C, T, b, b2 of a spectrum used to run
// through ONE accumulator chain over the whole pixel axis -- 125 tiles x 6 products = 750 instructions at c3 -- and the bias added
// up to ~4e-6 of the sums: the F gradient of the bench's 100 000 spectra came out 1.9e-4 from the float64 oracle (cancellation 47x),
// 3.2e-5 when the SAME spectra ran in launches small enough for the work plan to cut the pixel axis into 25-tile segments
// (tools/c3_100k_vs_oracle.py, tools/sections_vs_chunks.py).  Now the six products of a tile start from C = 0 and their sum joins
// the running sum by a float32 VALU add (round to nearest, no bias): the chain is six instructions long whatever N_pix is.
// (N_h = 17..32 keeps its chains, cut at QFA_P1_MAX_CHAIN tiles by the work plan: its accumulators live in the AGPR half of the
// file and the extra moves collide with the register prefetches.)
#ifndef QFA_P1_FRESH
#define QFA_P1_FRESH 1
#endif
__device__ __forceinline__ void acc_add(f32x4 &acc, const f32x4 &t) {
    typedef float f32x2p __attribute__((ext_vector_type(2)));            // two v_pk_add_f32 instead of four v_add_f32
    const f32x2p lo = f32x2p{acc[0], acc[1]} + f32x2p{t[0], t[1]}, hi = f32x2p{acc[2], acc[3]} + f32x2p{t[2], t[3]};
    acc = f32x4{lo[0], lo[1], hi[0], hi[1]};
}
#ifndef QFA_P1_SKIPT_KP
#define QFA_P1_SKIPT_KP 16     // prediction leaves out the T-side moments from this KP on (8: measured equal to 2.5 % slower, with or without an occupancy cap)
#endif
#ifndef QFA_P1_PIPE
#define QFA_P1_PIPE 1          // MFMA phase of pass 1 (N_h <= 16) as an explicit pipeline over the column tiles (mfmas_pipe)
#endif
#ifndef QFA_P1_PIPE_RED
#define QFA_P1_PIPE_RED 1      // groups requested ahead on a red tile (a group = two column tiles, 12 MFMAs) ...
#endif
#ifndef QFA_P1_PIPE_BLUE
#define QFA_P1_PIPE_BLUE 1     // ... and on a blue tile (a group = one column tile, 12 MFMAs); 2 before the fresh accumulators of round 5
                               // took eight registers
#endif
#ifndef QFA_P1_PIPE_BLUE_ZABS
#define QFA_P1_PIPE_BLUE_ZABS 0
#endif
#ifndef QFA_P1_MULMASK
#define QFA_P1_MULMASK 1       // pass 1: the pixel mask as a float factor instead of selects / exec-mask branches (weights())
#endif
#ifndef QFA_P1_STAMPS
#define QFA_P1_STAMPS 0    // diagnostic build (tools/p1_stamps.sh): s_memtime shares of the tile steps of one wave of pass 1
#endif
#if QFA_P1_STAMPS
__device__ unsigned long long qfa_p1_stamps[2 * 16];
#define P1S(i)                                                                                 \
    {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        st_[(BLUE ? 8 : 0) + i] += (unsigned)(t_ - st_last);                                   \
        st_last = t_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
#else
#define P1S(i) {}
#endif
template <int KP, bool PREDICT, int NW, bool ZF>      // ZF: factored-z input form (ZS = per-spectrum factors; zabs is not read)
__global__ __launch_bounds__(64 * NW, (KP > 16 || NW == 8) ? 1 : 2) void k_moments_x(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau,
                                                      const float *__restrict__ mu, int B, int Bpad, int Npix, int Nb,
                                                      int ntiles, WorkPlan wp, const unsigned char *__restrict__ PFX,
                                                      const float4 *__restrict__ ZS, float *__restrict__ MOM) {
    using C = Cfg<KP>;
    using X = XCfg<KP>;
    static_assert(NW == 4 || (NW == 8 && X::NSW == 1), "8 waves: the two-group form of N_h <= 16");
    constexpr int RING = NW == 8 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING][X::SLOT_B];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = wave_uniform(tid >> 6);
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int s0 = (blk * NW + wv) * 16;
    const bool active = s0 < B;                                   // wave-uniform
    const int nbt = (Nb + 31) >> 5;                               // tiles that contain blue pixels
    const DevConsts k = load_consts(p, tau);
    const int sl = lane & 15, g = lane >> 4;
    const bool svalid = (s0 + sl) < B;
    const int srow = active ? min(sl, B - 1 - s0) : 0;
    // the lane's spectrum is row `rowi` of the batch arrays (qfa_common.h, batch_row): per-lane 64-bit element offsets of
    // its 8 pixels of tile 0; the wave-uniform part of an address is array base + 32 * tile (SGPRs)
    const unsigned long long rowi = batch_row(bt, (active ? s0 : 0) + srow);
    const unsigned long long eoN = rowi * (unsigned long long)bt.row_stride + (unsigned)(8 * g);
    const unsigned long long eoB = ZF ? 0ull : rowi * (unsigned long long)(unsigned)Nb + (unsigned)(8 * g);
    const float *abase = bt.A_blue ? bt.A_blue + (size_t)(active ? s0 : 0) * Nb : nullptr;      // (batch order: not indexed)
    const int offB = srow * Nb;
    const ZFac zs = zfac_load(ZS, s0 + sl, ZF && svalid);

#if QFA_P1_STAMPS
    unsigned st_[16];
    for (int i = 0; i < 16; ++i) st_[i] = 0;
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    f32x4 accC[C::NT], accT[C::NT], accb[C::NFT], accb2[C::NFT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) accC[t] = accT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < C::NFT; ++t) accb[t] = accb2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    double qd = 0.0, ld = 0.0;        // float32 inside an 8-pixel group, float64 across groups
    float cn = 0.f, cblue = 0.f;

    // LDS-DMA of one image tile: wave w moves the 1-KiB pieces w, w+4, ...
    // (sub-image j of tile tg into ring slot buf; NSW = 1: the whole tile)
    auto stage_sub = [&](int tg, auto jtag, int buf) {
        constexpr int J = decltype(jtag)::value;
        constexpr int NCH = X::sub_bytes(J) / 1024, LO = NCH / NW, EX = NCH % NW;
        // each wave a contiguous run of pieces behind one write of M0 per five (glds16_run); waves below EX one piece more
        const int first = wv * LO + min(wv, EX);
        const unsigned char *src = PFX + (size_t)tg * X::TILE_B + X::sub_off(J) + first * 1024;
        const unsigned d = wave_uniform(lds_addr(&lds[buf][first * 1024]));
        if (EX && wv < EX) glds16_runs<LO + 1>(src, (unsigned)lane * 16u, d);
        else glds16_runs<LO>(src, (unsigned)lane * 16u, d);
    };
    auto stage = [&](int tg, int buf) { stage_sub(tg, std::integral_constant<int, 0>{}, buf); };

    auto run = [&](auto blue_tag, int ta, int tb) {
        constexpr bool BLUE = decltype(blue_tag)::value;
        // the T-side moments (T, b2: weights wD A^3, wD A^2 d) feed the gradients only -- the prediction instantiation
        // (k_solve<KP, true> reads C, b and the scalars) leaves them out: half the MFMAs and two of four splits on a blue tile
        // (N_h <= 8 keeps them: the tile step is VALU-bound there and without them pass 1 measured 2.5 % slower, DESI shape)
        constexpr bool TSIDE = BLUE && !(PREDICT && KP >= QFA_P1_SKIPT_KP);
        const int n = tb - ta;
        if (n <= 0) return;                                       // block-uniform

        // returns true when the loads were issued as (untracked) asm loads
        auto load_spec = [&](int tg, SpecRegsX &r) {
            const int pb = 32 * tg + 8 * g;
            if (!QFA_TRACKED_LOADS && pb + 7 < Npix) {
                const unsigned char *pd = lane_ptr<2>(bt.delta + 32 * tg, eoN), *pe = lane_ptr<2>(bt.error + 32 * tg, eoN);
                aload16p<0>(r.d0, pd);
                aload16p<16>(r.d1, pd);
                aload16p<0>(r.s0, pe);
                aload16p<16>(r.s1, pe);
                aload8p(r.m, lane_ptr<0>(bt.mask + 32 * tg, eoN));
            } else {                                              // ragged end of the pixel axis (ordinary loads)
                unsigned m0 = 0, m1 = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned long long o = eoN + (unsigned long long)(long long)(min(pb + e, Npix - 1) - 8 * g);
                    const float dv = bt.delta[o], sv = bt.error[o];
                    if (e < 4) { r.d0[e] = dv; r.s0[e] = sv; }
                    else { r.d1[e - 4] = dv; r.s1[e - 4] = sv; }
                    const unsigned bit = (pb + e < Npix && bt.mask[o] != 0) ? (1u << (8 * (e & 3))) : 0u;
                    if (e < 4) m0 |= bit;
                    else m1 |= bit;
                }
                r.m[0] = m0;
                r.m[1] = m1;
                // retire these (tracked) loads here: pending at the join, they would make the compiler wait
                // vmcnt(0) in front of the fast path's next loads into the same registers
                asm volatile("" : "+v"(r.d0), "+v"(r.d1), "+v"(r.s0), "+v"(r.s1));
            }
            if (BLUE && !ZF) {
                if (!QFA_TRACKED_LOADS && pb + 7 < Nb) {
                    const unsigned char *pz = lane_ptr<2>(bt.zabs + 32 * tg, eoB);
                    aload16p<0>(r.z0, pz);
                    aload16p<16>(r.z1, pz);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float zv = bt.zabs[eoB + (unsigned long long)(long long)(min(pb + e, Nb - 1) - 8 * g)];
                        if (e < 4) r.z0[e] = zv;
                        else r.z1[e - 4] = zv;
                    }
                    asm volatile("" : "+v"(r.z0), "+v"(r.z1));
                }
            }
        };

        // ---- phase 1 of a tile: per-element weights on the VALU (QFA/model.py:125-131), split into bf16 pieces
        auto weights = [&](int tg, const SpecRegsX &cur, const unsigned char *rows, Pieces &w) {
            const float *pp = reinterpret_cast<const float *>(rows) + 8 * g;      // the tile's parameter rows (OFF_PSI of its image)
            float psi[8], om[8], muv[8], ti[8], pwi[8], ofl[8], bluef[8], swt[8];
            if (X::F16) {           // s_px of the C / T weights (row 7 of the tile's parameters)
                const float4 a = *reinterpret_cast<const float4 *>(pp + 224), b = *reinterpret_cast<const float4 *>(pp + 228);
                swt[0] = a.x; swt[1] = a.y; swt[2] = a.z; swt[3] = a.w; swt[4] = b.x; swt[5] = b.y; swt[6] = b.z; swt[7] = b.w;
            }
            if (BLUE && ZF) {
                const float4 a = *reinterpret_cast<const float4 *>(pp + 96), b = *reinterpret_cast<const float4 *>(pp + 100),
                             c = *reinterpret_cast<const float4 *>(pp + 128), d = *reinterpret_cast<const float4 *>(pp + 132),
                             e = *reinterpret_cast<const float4 *>(pp + 160), f = *reinterpret_cast<const float4 *>(pp + 164);
                ti[0] = a.x; ti[1] = a.y; ti[2] = a.z; ti[3] = a.w; ti[4] = b.x; ti[5] = b.y; ti[6] = b.z; ti[7] = b.w;
                pwi[0] = c.x; pwi[1] = c.y; pwi[2] = c.z; pwi[3] = c.w; pwi[4] = d.x; pwi[5] = d.y; pwi[6] = d.z; pwi[7] = d.w;
                ofl[0] = e.x; ofl[1] = e.y; ofl[2] = e.z; ofl[3] = e.w; ofl[4] = f.x; ofl[5] = f.y; ofl[6] = f.z; ofl[7] = f.w;
            }
            if (BLUE) {
                const float4 a = *reinterpret_cast<const float4 *>(pp + 192), b = *reinterpret_cast<const float4 *>(pp + 196);
                bluef[0] = a.x; bluef[1] = a.y; bluef[2] = a.z; bluef[3] = a.w;
                bluef[4] = b.x; bluef[5] = b.y; bluef[6] = b.z; bluef[7] = b.w;
            }
            if (PREDICT) {          // from the tile image: a global load here would sit in the counted vmcnt queue
                const float4 a = *reinterpret_cast<const float4 *>(pp + 64), b = *reinterpret_cast<const float4 *>(pp + 68);
                muv[0] = a.x; muv[1] = a.y; muv[2] = a.z; muv[3] = a.w;
                muv[4] = b.x; muv[5] = b.y; muv[6] = b.z; muv[7] = b.w;
            }
            {
                const float4 a = *reinterpret_cast<const float4 *>(pp), b = *reinterpret_cast<const float4 *>(pp + 4);
                psi[0] = a.x; psi[1] = a.y; psi[2] = a.z; psi[3] = a.w;
                psi[4] = b.x; psi[5] = b.y; psi[6] = b.z; psi[7] = b.w;
                if (BLUE) {
                    const float4 c = *reinterpret_cast<const float4 *>(pp + 32),
                                 d = *reinterpret_cast<const float4 *>(pp + 36);
                    om[0] = c.x; om[1] = c.y; om[2] = c.z; om[3] = c.w;
                    om[4] = d.x; om[5] = d.y; om[6] = d.z; om[7] = d.w;
                }
            }
            float qd8 = 0.f, ld8 = 0.f;
#if QFA_P1_MULMASK
            // The mask as a FACTOR (round 4).  `w ? x : 0` per pixel compiles to an exec-mask branch per pixel (saveexec / xor /
            // or, zero-initialising moves: ~100 of the 440 instructions of a red tile step) or to three selects.  Here the
            // mask byte becomes m = 0.0 / 1.0 with ONE instruction (v_cvt_f32_ubyteN; min(., 1) so that any nonzero byte
            // counts) and multiplies 1/D, log D and the count; D and delta of a masked pixel are clamped to finite values
            // first (min / max drop a NaN: a masked pixel may hold anything, not only the reference's -999), so m = 0 times
            // them is 0.  For unmasked pixels every product is by exactly 1: the same bits as the select form.
            const unsigned mw0 = svalid ? cur.m[0] : 0u, mw1 = svalid ? cur.m[1] : 0u;
            // D is clamped from BELOW too: a pad pixel of the ragged last tile has Psi = 0 in the image and clones
            // sigma of pixel Npix - 1, a masked pixel may carry error = 0 under Psi = 0 -- D = 0, 1/D = inf, and 0 x inf = NaN
            // would reach the MFMA operands and the NLL of the whole spectrum.  One v_med3 in the place of the v_min (a NaN
            // makes it return the smallest operand: finite as well).
            constexpr float BIG = 3.0e38f, TINY = 1.0e-37f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float c2[2], c3[2], cb[2], cb2[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int e = 2 * q + h2;
                    const int px = 32 * tg + 8 * g + e;
                    const unsigned mw = e < 4 ? mw0 : mw1;
                    float mf = (float)((mw >> (8 * (e & 3))) & 0xffu);       // (hipcc selects v_cvt_f32_ubyteN for this)
                    mf = fminf(fmaxf(mf, 0.f), 1.f);                         // (the clamp bit of that instruction: no op of its own)
                    float d = e < 4 ? cur.d0[e & 3] : cur.d1[e & 3];
                    const float sg = e < 4 ? cur.s0[e & 3] : cur.s1[e & 3];
                    d = __builtin_amdgcn_fmed3f(d, -BIG, BIG);              // (v_med3_f32 returns min3 when an operand is NaN: -BIG)
                    float D, wD;
                    if (BLUE) {
                        float A, zdom;
                        if (ZF) {
                            // factored-z form (never with a custom A_blue): a red pixel of the tile across the boundary has
                            // ti = pwi = omega = offset 0 in the image -- A = exp2(0) = 1 and omega zd = 0 without a select
                            const float pw = zs.pw * pwi[e];                                      // utils.py:73
                            A = fast_exp2(fmaf(zs.ts, ti[e], ofl[e]));                            // QFA/model.py:125
                            const float re = k.omc0 - fast_exp2(k.k1 * pw);                       // utils.py:91
                            zdom = re * re * om[e];
                        } else {
                            const bool blue = px < Nb;
                            const BlueTerms t = blue_terms(e < 4 ? cur.z0[e & 3] : cur.z1[e & 3], k);
                            float Ab = t.A;
                            if (abase) Ab = abase[offB + min(px, Nb - 1)];    // custom tau callable (rare path)
                            A = blue ? Ab : 1.f;
                            zdom = blue ? t.zd * om[e] : 0.f;
                        }
                        D = __builtin_amdgcn_fmed3f(A * A * psi[e] + zdom + sg * sg, TINY, BIG);
                        if (PREDICT) d = d - muv[e] * A;                     // QFA/model.py:166
                        wD = mf * fast_rcp(D);
                        const float wDA = wD * A;
                        c2[h2] = wDA * A;
                        cb[h2] = wDA * d;
                        if (TSIDE) cb2[h2] = c2[h2] * d;
                        if (X::F16) c2[h2] *= swt[e];                        // (<= 2^12: wD A^2 <= 1 / Psi)
                        if (TSIDE) c3[h2] = c2[h2] * A;
                        cblue = fmaf(mf, bluef[e], cblue);
                    } else {                                                 // red side: A = 1, no omega term
                        D = __builtin_amdgcn_fmed3f(psi[e] + sg * sg, TINY, BIG);
                        if (PREDICT) d = d - muv[e];
                        wD = mf * fast_rcp(D);
                        c2[h2] = X::F16 ? wD * swt[e] : wD;
                        cb[h2] = wD * d;
                    }
                    qd8 += wD * d * d;
                    ld8 = fmaf(mf, fast_log2(D), ld8);                       // (log 2 once per tile, below)
                    cn += mf;
                }
                unsigned h, m, l;
                if (X::F16) { split2h(c2[0], c2[1], h, m); w.w1h[q] = h; w.w1m[q] = m; w.w1l[q] = 0u; }
                else { split2(c2[0], c2[1], h, m, l); w.w1h[q] = h; w.w1m[q] = m; w.w1l[q] = l; }
                split2(cb[0], cb[1], h, m, l);
                w.w3h[q] = h; w.w3m[q] = m; w.w3l[q] = l;
                if (TSIDE) {
                    if (X::F16) { split2h(c3[0], c3[1], h, m); w.w2h[q] = h; w.w2m[q] = m; w.w2l[q] = 0u; }
                    else { split2(c3[0], c3[1], h, m, l); w.w2h[q] = h; w.w2m[q] = m; w.w2l[q] = l; }
                    split2(cb2[0], cb2[1], h, m, l);
                    w.w4h[q] = h; w.w4m[q] = m; w.w4l[q] = l;
                }
                pin(qd8, ld8);
                __builtin_amdgcn_sched_barrier(0);
            }
            qd += (double)qd8;
            ld += (double)(ld8 * QFA_LN2);
            return;
#endif
            // pixel pair by pixel pair: weights of two pixels, then their bf16 pieces (short live ranges)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float c2[2], c3[2], cb[2], cb2[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int e = 2 * q + h2;
                    const int px = 32 * tg + 8 * g + e;
                    const unsigned mb = (cur.m[e >> 2] >> (8 * (e & 3))) & 0xffu;
                    const bool wv_ = svalid & (mb != 0);
                    float d = e < 4 ? cur.d0[e & 3] : cur.d1[e & 3];
                    const float sg = e < 4 ? cur.s0[e & 3] : cur.s1[e & 3];
                    float D, wD;
                    if (BLUE) {
                        const bool blue = px < Nb;
                        const BlueTerms t = ZF ? blue_terms_zf(zs, ti[e], pwi[e], 0.f, k)
                                               : blue_terms(e < 4 ? cur.z0[e & 3] : cur.z1[e & 3], k);
                        float Ab = t.A;
                        if (abase) Ab = abase[offB + min(px, Nb - 1)];        // custom tau callable (rare path)
                        const float A = blue ? Ab : 1.f;
                        const float zdom = blue ? t.zd * om[e] : 0.f;
                        D = A * A * psi[e] + zdom + sg * sg;
                        if (PREDICT) d = d - muv[e] * A;                     // QFA/model.py:166
                        wD = wv_ ? fast_rcp(D) : 0.f;
                        d = wv_ ? d : 0.f;
                        const float wDA = wD * A;
                        c2[h2] = wDA * A;
                        cb[h2] = wDA * d;
                        if (TSIDE) cb2[h2] = c2[h2] * d;
                        if (X::F16) c2[h2] *= swt[e];
                        if (TSIDE) c3[h2] = c2[h2] * A;
                        cblue += (wv_ & blue) ? 1.f : 0.f;
                    } else {                                                 // red side: A = 1, no omega term
                        D = psi[e] + sg * sg;
                        if (PREDICT) d = d - muv[e];
                        wD = wv_ ? fast_rcp(D) : 0.f;
                        d = wv_ ? d : 0.f;
                        c2[h2] = X::F16 ? wD * swt[e] : wD;
                        cb[h2] = wD * d;
                    }
                    qd8 += wD * d * d;
                    ld8 += wv_ ? fast_log(D) : 0.f;
                    cn += wv_ ? 1.f : 0.f;
                }
                unsigned h, m, l;
                if (X::F16) { split2h(c2[0], c2[1], h, m); w.w1h[q] = h; w.w1m[q] = m; w.w1l[q] = 0u; }
                else { split2(c2[0], c2[1], h, m, l); w.w1h[q] = h; w.w1m[q] = m; w.w1l[q] = l; }
                split2(cb[0], cb[1], h, m, l);
                w.w3h[q] = h; w.w3m[q] = m; w.w3l[q] = l;
                if (TSIDE) {
                    if (X::F16) { split2h(c3[0], c3[1], h, m); w.w2h[q] = h; w.w2m[q] = m; w.w2l[q] = 0u; }
                    else { split2(c3[0], c3[1], h, m, l); w.w2h[q] = h; w.w2m[q] = m; w.w2l[q] = l; }
                    split2(cb2[0], cb2[1], h, m, l);
                    w.w4h[q] = h; w.w4m[q] = m; w.w4l[q] = l;
                }
                pin(qd8, ld8);
                __builtin_amdgcn_sched_barrier(0);
            }
            qd += (double)qd8;
            ld += (double)ld8;
        };
        // ---- phase 3: per 16-column tile of the image three ds_read_b128, then 6 (red) or 12 (blue) XDL MFMAs.
        // J: the sub-image in `tile` (N_h > 16: column tiles [0, CT1) or [CT1, NCT); else all of them)
        auto mfmas = [&](const unsigned char *tile, const Pieces &w, auto jtag) {
            constexpr int J = decltype(jtag)::value;
            constexpr int CT0 = J == 0 ? 0 : X::CT1, CTE = J == 0 ? X::CT1 : X::NCT, PS = X::pstr(J);
            const unsigned char *bcol = tile + lane * 16;            // lane-linear: conflict-free ds_read_b128
            auto rdB = [&](int piece, int ct) {
                return *reinterpret_cast<const u32x4 *>(bcol + piece * PS + (ct - CT0) * 1024);
            };
            // (XCfg::F16: the pair column tiles are two float16 pieces -- three products)
            auto ct_terms = [&](const u32x4 &ah, const u32x4 &am, const u32x4 &al, int ct, f32x4 c) __attribute__((always_inline)) {
                if constexpr (X::F16) return xdl3h(ah, am, rdB(0, ct), rdB(1, ct), c);
                else return xdl_ct(ah, am, al, rdB(0, ct), rdB(1, ct), rdB(2, ct), c);
            };
#pragma unroll
            for (int t = 0; t < C::NFT; ++t) {
                if (t >= CT0 && t < CTE) {
                    const u32x4 bh = rdB(0, t), bm = rdB(1, t), bl = rdB(2, t);
                    if (QFA_P1_FRESH && KP <= 16) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        acc_add(accb[t], xdl6(w.w3h, w.w3m, w.w3l, bh, bm, bl, z));
                        if (TSIDE) acc_add(accb2[t], xdl6(w.w4h, w.w4m, w.w4l, bh, bm, bl, z));
                    } else {
                    accb[t] = xdl6(w.w3h, w.w3m, w.w3l, bh, bm, bl, accb[t]);
                    if (TSIDE) accb2[t] = xdl6(w.w4h, w.w4m, w.w4l, bh, bm, bl, accb2[t]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                if (C::NFT + t >= CT0 && C::NFT + t < CTE) {
                    if (QFA_P1_FRESH && KP <= 16) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        acc_add(accC[t], ct_terms(w.w1h, w.w1m, w.w1l, C::NFT + t, z));
                        if (TSIDE) acc_add(accT[t], ct_terms(w.w2h, w.w2m, w.w2l, C::NFT + t, z));
                    } else {
                    accC[t] = ct_terms(w.w1h, w.w1m, w.w1l, C::NFT + t, accC[t]);
                    if (TSIDE) accT[t] = ct_terms(w.w2h, w.w2m, w.w2l, C::NFT + t, accT[t]);
                    }
                }
            }
        };

        // The same phase for a whole tile in LDS (N_h <= 16) as an explicit software pipeline: the B operand of column
        // tile t + D - 1 is requested before the MFMAs of column tile t are issued (hipcc's own order asks for a column
        // tile right in front of its first MFMA: ~100 cycles of LDS latency in the open per column tile, 2 090 cycles for
        // 960 of XDL work on a red tile), and the requests of the step -- the image tile c + 1 piece by piece, then the
        // spectra of tile c + 2 -- are issued between the column tiles, in the issue slots the MFMAs leave free.
        auto mfmas_pipe = [&](const unsigned char *tile, const Pieces &w, bool do_stage, int tg_next, int buf_next,
                              bool reload, int tg_spec, SpecRegsX &cur) __attribute__((always_inline)) {
            // column tiles in groups of GS whose MFMA chains alternate (a chain of six on one accumulator issues every
            // ~19 cycles, two alternating chains every 16): red tiles pair two column tiles, blue tiles the C-side and
            // the T-side chain of one; PFG groups are requested ahead
            // (the zabs instantiation -- two more spectra registers sets of 8 -- requests its blue operands one group later: it has no
            // registers for the two fresh accumulators otherwise)
            constexpr int NCT = X::NCT, PS = X::pstr(0), GS = TSIDE ? 1 : 2, PFG = TSIDE ? (ZF ? QFA_P1_PIPE_BLUE : QFA_P1_PIPE_BLUE_ZABS) : QFA_P1_PIPE_RED;
            constexpr int D = GS * (PFG + 1), NG = (NCT + GS - 1) / GS;
            constexpr int NR = (X::NCHUNK + NW - 1) / NW;                        // DMA requests per wave
            constexpr int RPG = (NR + NG - 2) / (NG - 1);                        // ... per group
            const unsigned char *bcol = tile + lane * 16;
            u32x4 bh[D], bm[D], bl[D];
            auto rd = [&](int ct) __attribute__((always_inline)) {
                bh[ct % D] = *reinterpret_cast<const u32x4 *>(bcol + ct * 1024);
                bm[ct % D] = *reinterpret_cast<const u32x4 *>(bcol + PS + ct * 1024);
                if (!X::F16 || ct < C::NFT) bl[ct % D] = *reinterpret_cast<const u32x4 *>(bcol + 2 * PS + ct * 1024);
            };
            auto acc_of = [&](int t, bool second) -> f32x4 & {
                return t < C::NFT ? (second ? accb2[t] : accb[t]) : (second ? accT[t - C::NFT] : accC[t - C::NFT]);
            };
            // six_terms' order, two chains alternating
            // (QFA_P1_FRESH: c0 / c1 are the group's two FRESH accumulators tm0 / tm1; their sums join the running sums behind the
            // group's requests -- by then the MFMAs have delivered -- see the loop below)
            f32x4 tm0 = {0.f, 0.f, 0.f, 0.f}, tm1 = {0.f, 0.f, 0.f, 0.f};
            auto six2 = [&](const u32x4 &ah, const u32x4 &am, const u32x4 &al, int q0, f32x4 &c0, const u32x4 &eh,
                            const u32x4 &em, const u32x4 &el, int q1, f32x4 &c1) __attribute__((always_inline)) {
                c0 = xdl(ah, bl[q0], c0); c1 = xdl(eh, bl[q1], c1);
                c0 = xdl(al, bh[q0], c0); c1 = xdl(el, bh[q1], c1);
                c0 = xdl(am, bm[q0], c0); c1 = xdl(em, bm[q1], c1);
                c0 = xdl(am, bh[q0], c0); c1 = xdl(em, bh[q1], c1);
                c0 = xdl(ah, bm[q0], c0); c1 = xdl(eh, bm[q1], c1);
                c0 = xdl(ah, bh[q0], c0); c1 = xdl(eh, bh[q1], c1);
            };
            // XCfg::F16: the chains of a PAIR column tile are three float16 products (xdl3h's order); the F tile keeps six bf16 ones
            auto three2 = [&](const u32x4 &ah, const u32x4 &am, int q0, f32x4 &c0, const u32x4 &eh, const u32x4 &em, int q1,
                              f32x4 &c1) __attribute__((always_inline)) {
                c0 = xdlh(ah, bm[q0], c0); c1 = xdlh(eh, bm[q1], c1);
                c0 = xdlh(am, bh[q0], c0); c1 = xdlh(em, bh[q1], c1);
                c0 = xdlh(ah, bh[q0], c0); c1 = xdlh(eh, bh[q1], c1);
            };
            // red tile, group 0: the F tile (six bf16 products, chain c0) beside the first pair tile (three float16 products, c1)
            auto six_three = [&](const u32x4 &ah, const u32x4 &am, const u32x4 &al, int q0, f32x4 &c0, const u32x4 &eh,
                                 const u32x4 &em, int q1, f32x4 &c1) __attribute__((always_inline)) {
                c0 = xdl(ah, bl[q0], c0); c1 = xdlh(eh, bm[q1], c1);
                c0 = xdl(al, bh[q0], c0); c1 = xdlh(em, bh[q1], c1);
                c0 = xdl(am, bm[q0], c0); c1 = xdlh(eh, bh[q1], c1);
                c0 = xdl(am, bh[q0], c0);
                c0 = xdl(ah, bm[q0], c0);
                c0 = xdl(ah, bh[q0], c0);
            };
#pragma unroll
            for (int t = 0; t < GS * PFG; ++t)
                if (t < NCT) rd(t);
            bool spec_done = false;
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
                for (int j = 0; j < GS; ++j)
                    if (GS * (gi + PFG) + j < NCT) rd(GS * (gi + PFG) + j);
                __builtin_amdgcn_sched_barrier(0);
                const int t = GS * gi;
                // the group's two accumulators: (C side, T side) of column tile t on a blue tile, column tiles t and t + 1 on a red one
                f32x4 &ga0 = acc_of(t, false), &ga1 = TSIDE ? acc_of(t, true) : acc_of(t + 1 < NCT ? t + 1 : t, false);
                const bool two = TSIDE || t + 1 < NCT;
                if (QFA_P1_FRESH) tm0 = tm1 = f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 &c0 = QFA_P1_FRESH ? tm0 : ga0, &c1 = QFA_P1_FRESH ? tm1 : ga1;
                if constexpr (X::F16) {
                    static_assert(!X::F16 || C::NFT == 1, "one F tile in front of the pair tiles");
                    if (TSIDE) {
                        if (t < C::NFT) six2(w.w3h, w.w3m, w.w3l, t % D, c0, w.w4h, w.w4m, w.w4l, t % D, c1);
                        else three2(w.w1h, w.w1m, t % D, c0, w.w2h, w.w2m, t % D, c1);
                    } else if (t + 1 < NCT) {
                        if (t < C::NFT) six_three(w.w3h, w.w3m, w.w3l, t % D, c0, w.w1h, w.w1m, (t + 1) % D, c1);
                        else three2(w.w1h, w.w1m, t % D, c0, w.w1h, w.w1m, (t + 1) % D, c1);
                    } else {
                        c0 = t < C::NFT ? xdl6(w.w3h, w.w3m, w.w3l, bh[t % D], bm[t % D], bl[t % D], c0)
                                        : xdl3h(w.w1h, w.w1m, bh[t % D], bm[t % D], c0);
                    }
                } else if (TSIDE) {
                    if (t < C::NFT) six2(w.w3h, w.w3m, w.w3l, t % D, c0, w.w4h, w.w4m, w.w4l, t % D, c1);
                    else six2(w.w1h, w.w1m, w.w1l, t % D, c0, w.w2h, w.w2m, w.w2l, t % D, c1);
                } else if (t + 1 < NCT) {
                    const bool f0 = t < C::NFT, f1 = t + 1 < C::NFT;
                    six2(f0 ? w.w3h : w.w1h, f0 ? w.w3m : w.w1m, f0 ? w.w3l : w.w1l, t % D, c0,
                         f1 ? w.w3h : w.w1h, f1 ? w.w3m : w.w1m, f1 ? w.w3l : w.w1l, (t + 1) % D, c1);
                } else {
                    c0 = t < C::NFT ? xdl6(w.w3h, w.w3m, w.w3l, bh[t % D], bm[t % D], bl[t % D], c0)
                                    : xdl6(w.w1h, w.w1m, w.w1l, bh[t % D], bm[t % D], bl[t % D], c0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = gi * RPG; i < (gi + 1) * RPG; ++i) {
                    if (i < NR && do_stage) {                    // the wave's pieces: the contiguous run stage_sub moves
                        constexpr int LO = X::NCHUNK / NW, EX = X::NCHUNK % NW;
                        const int ch = wv * LO + min(wv, EX) + i;
                        if (i < LO || wv < EX)
                            glds16a_nc(PFX + (size_t)tg_next * X::TILE_B + ch * 1024, (unsigned)lane * 16u,
                                       wave_uniform(lds_addr(&lds[buf_next][ch * 1024])));
                    }
                }
                if (!spec_done && (gi + 1) * RPG >= NR) {
                    spec_done = true;
                    if (reload) load_spec(tg_spec, cur);
                }
                if (QFA_P1_FRESH) {
                    acc_add(ga0, tm0);
                    if (two) acc_add(ga1, tm1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // One tile.  Phase 1 consumes the spectra registers of tile c; phase 2 starts the LDS-DMA of image tile
        // c+1 and reloads the same registers with tile c+2 (those loads have the MFMA phase of this step and the
        // whole next step to land, so HBM requests are in flight all the time); phase 3 issues the MFMAs.  The
        // raw barrier waits for the DMA only: vmcnt counts in issue order and the spectra loads come after it.
        auto step = [&](int c, SpecRegsX &cur, int buf) {
            Pieces w;
            if (QFA_P1_EARLY_DMA && c + 1 < n && !(QFA_P1_ABL & 8)) stage(ta + c + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            P1S(0)
            land<BLUE && !ZF>(cur);
            P1S(1)
            if (QFA_P1_ABL & 4) {            // timing only: no weights (pieces straight from the spectra registers)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w.w1h[q] = w.w1m[q] = w.w1l[q] = __float_as_uint(cur.d0[q]);
                    w.w3h[q] = w.w3m[q] = w.w3l[q] = __float_as_uint(cur.s0[q]);
                    w.w2h[q] = w.w2m[q] = w.w2l[q] = __float_as_uint(cur.d1[q]);
                    w.w4h[q] = w.w4m[q] = w.w4l[q] = __float_as_uint(cur.s1[q]);
                }
            } else if (active) weights(ta + c, cur, lds[buf] + X::OFF_PSI, w);
            __builtin_amdgcn_sched_barrier(0);
            P1S(2)
            const bool reload = active & (c + 2 < n) & !(QFA_P1_ABL & 1);
            if (QFA_P1_PIPE && KP == 16 && QFA_P1_CT_TERMS == 6 && !QFA_P1_ABL && !QFA_P1_EARLY_DMA) {
                if (active) mfmas_pipe(lds[buf], w, c + 1 < n, ta + c + 1, buf ^ 1, reload, ta + c + 2, cur);
                else if (c + 1 < n) stage(ta + c + 1, buf ^ 1);
            } else {
            if (!QFA_P1_EARLY_DMA && c + 1 < n && !(QFA_P1_ABL & 8)) stage(ta + c + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            P1S(7)
            if (reload) load_spec(ta + c + 2, cur);
            __builtin_amdgcn_sched_barrier(0);
            P1S(3)
            if (QFA_P1_ABL & 2) {            // timing only: no MFMAs (the pieces stay live)
                asm volatile("" ::"v"(w.w1h), "v"(w.w1m), "v"(w.w1l), "v"(w.w3h), "v"(w.w3m), "v"(w.w3l));
                if (TSIDE) asm volatile("" ::"v"(w.w2h), "v"(w.w2m), "v"(w.w2l), "v"(w.w4h), "v"(w.w4m), "v"(w.w4l));
            } else if (active) mfmas(lds[buf], w, std::integral_constant<int, 0>{});
            }
            P1S(4)
            // retire everything up to and including the DMA: it was issued before the 5 (red: 2 delta, 2 sigma,
            // 1 mask) / 7 (blue: + 2 zabs) spectra loads of this step (the ragged-end path issues more, smaller ones)
            if (reload) dma_wait<(BLUE && !ZF) ? 7 : 5>();
            else dma_wait<0>();
            P1S(5)
            wg_barrier();
            asm volatile("" ::: "memory");
            P1S(6)
        };

        // N_h > 16, one tile in two sweeps: sub-image 0 lives in ring slot 0, sub-image 1 in slot 1.  Sweep 0: weights,
        // DMA of this tile's sub-image 1, the spectra of tile c + 2, the MFMAs of the first column tiles; sweep 1: DMA of
        // the next tile's sub-image 0 (slot 0 is free behind the barrier), the remaining MFMAs.  The second wait is for
        // everything: the spectra requested in sweep 0 have had a whole tile step (~3 us at N_h = 32) by then.
        auto step2 = [&](int c, SpecRegsX &cur) {
            Pieces w;
            land<BLUE && !ZF>(cur);
            if (active) weights(ta + c, cur, lds[0] + X::OFF_PSI, w);
            __builtin_amdgcn_sched_barrier(0);
            stage_sub(ta + c, std::integral_constant<int, X::NSW - 1>{}, 1);
            __builtin_amdgcn_sched_barrier(0);
            const bool reload = active & (c + 2 < n);
            if (reload) load_spec(ta + c + 2, cur);
            __builtin_amdgcn_sched_barrier(0);
            if (active) mfmas(lds[0], w, std::integral_constant<int, 0>{});
            if (reload) dma_wait<(BLUE && !ZF) ? 7 : 5>();
            else dma_wait<0>();
            wg_barrier();
            asm volatile("" ::: "memory");
            if (c + 1 < n) stage(ta + c + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (active) mfmas(lds[1], w, std::integral_constant<int, X::NSW - 1>{});
            dma_wait<0>();
            wg_barrier();
            asm volatile("" ::: "memory");
        };

        // NW = 8 (N_h <= 16): ONE workgroup of 8 waves = 128 spectra per CU shares the image ring (half the LDS-DMA per
        // spectrum), waves w and w + 4 share a SIMD.  Group A (waves 0..3) runs [weights c | MFMAs c] in step c, group B
        // (waves 4..7) [MFMAs c - 1 | weights c]: the VALU phase of one beside the XDL phase of the other.  B reads image
        // c - 1 while image c + 1 arrives: ring of 3, the DMA issued at the start of the step (its slot was last read in
        // step c - 1, by B).
        const bool grpB = NW == 8 && wv >= 4;                    // wave-uniform
        Pieces wB;
        auto step8 = [&](int c, SpecRegsX &cur) {
            const int slot = c % 3, nslot = (c + 1) % 3, pslot = (c + 2) % 3;
            if (c + 1 < n) stage(ta + c + 1, nslot);
            __builtin_amdgcn_sched_barrier(0);
            land<BLUE && !ZF>(cur);
            const bool reload = active & (c + 2 < n);
            // (one weights site and one set of piece registers for both groups)
            if (grpB && active && c > 0) mfmas(lds[pslot], wB, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            if (active) weights(ta + c, cur, lds[slot] + X::OFF_PSI, wB);
            __builtin_amdgcn_sched_barrier(0);
            if (reload) load_spec(ta + c + 2, cur);
            __builtin_amdgcn_sched_barrier(0);
            if (!grpB && active) mfmas(lds[slot], wB, std::integral_constant<int, 0>{});
            if (reload) dma_wait<(BLUE && !ZF) ? 7 : 5>();
            else dma_wait<0>();
            wg_barrier();
            asm volatile("" ::: "memory");
        };

        SpecRegsX ra, rb;
        stage(ta, 0);
        if (active) {
            load_spec(ta, ra);
            if (n > 1) load_spec(ta + 1, rb);
        }
        dma_wait<0>();
        __syncthreads();
        if constexpr (NW == 8) {
            for (int c = 0; c < n; c += 2) {
                step8(c, ra);
                if (c + 1 < n) step8(c + 1, rb);
            }
            if (grpB && active) mfmas(lds[(n - 1) % 3], wB, std::integral_constant<int, 0>{});
        } else
        for (int c = 0; c < n; c += 2) {
            if constexpr (X::NSW == 1) {
                step(c, ra, 0);
                if (c + 1 < n) step(c + 1, rb, 1);
            } else {
                step2(c, ra);
                if (c + 1 < n) step2(c + 1, rb);
            }
        }
        __syncthreads();
    };
    run(std::false_type{}, max(t0, nbt), t1);
    constexpr bool SKIPT = PREDICT && KP >= QFA_P1_SKIPT_KP;
    if (!SKIPT) {
#pragma unroll
        for (int t = 0; t < C::NT; ++t) accT[t] = accC[t];
#pragma unroll
        for (int t = 0; t < C::NFT; ++t) accb2[t] = accb[t];
    }
    run(std::true_type{}, t0, min(t1, nbt));

#if QFA_P1_STAMPS
    if (blockIdx.x == 300 && wv == 0 && lane == 0)
        for (int i = 0; i < 16; ++i) qfa_p1_stamps[i] = st_[i];
#endif
    if (!active) return;
    // C/D layout: col = lane&15, row = 4*(lane>>4) + r  -> spectrum s0 + 4g + r, column 16t + sl
    if constexpr (X::F16) {            // the column scales of the pair part leave here (1 / cs: powers of two)
        const unsigned *cmx = pfx_colmax<KP>(const_cast<unsigned char *>(PFX), ntiles) + C::FW + sl;
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            float f;
            (void)pfx_col_scale(cmx[16 * t], f);
#pragma unroll
            for (int r = 0; r < 4; ++r) { accC[t][r] *= f; accT[t][r] *= f; }
        }
    }
    float *momseg = mom_segment<C::NMOM>(MOM, wp, seg, Bpad, 16 * NW);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ss = s0 + 4 * g + r;
        if (ss < B) {
            float *m = momseg + (size_t)ss * C::NMOM + sl;
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                m[16 * t] = accC[t][r];
                if (!SKIPT) m[C::MOM_T + 16 * t] = accT[t][r];         // (prediction, N_h > 8: T and b2 are neither computed nor read)
            }
#pragma unroll
            for (int t = 0; t < C::NFT; ++t) {
                m[C::MOM_B + 16 * t] = accb[t][r];
                if (!SKIPT) m[C::MOM_B2 + 16 * t] = accb2[t][r];
            }
        }
    }
    qd += __shfl_xor(qd, 16); qd += __shfl_xor(qd, 32);
    ld += __shfl_xor(ld, 16); ld += __shfl_xor(ld, 32);
    cn += __shfl_xor(cn, 16); cn += __shfl_xor(cn, 32);
    cblue += __shfl_xor(cblue, 16); cblue += __shfl_xor(cblue, 32);
    if (g == 0 && svalid) {
        float *m = momseg + (size_t)(s0 + sl) * C::NMOM + C::MOM_S;
        m[0] = (float)qd; m[1] = (float)ld; m[2] = cn; m[3] = cblue;
    }
}
