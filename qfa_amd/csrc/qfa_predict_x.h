// qfa_predict_x.h -- posterior writer for N_h <= 16 on the XDL pipe: cont = F hmean + mu, unc = sqrt(f^T hcov f)
// for ALL pixels of every spectrum (reference QFA/model.py:177-180).
//
// k_predict_out (qfa_step_kernels.h) issues [f^T hmean | f^T hcov f] as 38 v_mfma_f32_16x16x4_f32 per 16 x 16 outputs:
// 1 216 cycles of the SIMD's float32 datapath per 2 KiB written, i.e. compute-bound at 17-22 % of the HBM write rate
// (bench.py `predict`).  Here the same contraction is the six-term split-bf16 product of qfa_grads_x.h's stage 1
// (v_mfma_f32_16x16x32_bf16, both operands static: [hmean | hcov'] of 16 spectra in 72 / 36 VGPRs, the tile image
// [F^T | pair products] staged by LDS-DMA): 36 / 18 MFMAs of 16 cycles per 16 x 16 outputs.
//   lane (lo = lane & 15, g = lane >> 4) owns pixels 32 t + 2 lo + h (h = 0, 1) of the spectra s0 + 4 g + r: one 8-byte
//   store per spectrum row and array, 128 contiguous bytes per row and wave instruction.  A wave serves SPW groups of 16
//   spectra (template argument; the work plan's blocks are 64 SPW spectra, qfa_host.h).
#pragma once
#include "qfa_common.h"
#include "qfa_xdl_kernels.h"

#ifndef QFA_PX_F16
#define QFA_PX_F16 1
#endif
template <int KP>
struct PX {
    static constexpr int KK2 = KP * (KP + 1) / 2;
    static constexpr int NKS = 1 + (KK2 + 31) / 32;          // K-steps: [hmean, 0 | pair products]; 3 at KP = 8, 6 at 16
    // Round 5 (QFA_PX_F16): both operands as TWO float16 pieces, three products per K-step (qfa_common.h "float16 pieces").
    // The image holds t f_a and t^2 f_a f_b with t the pixel's power of two; 1 / t and 1 / t^2 sit as float32 in K-step 0's
    // h piece, lanes g = 3 (K slots 24..31: they meet the zeros behind hmean).  [hmean] and [hcov'] of a spectrum get powers
    // of two of their own where the kernel builds its A operand.
    static constexpr bool F16 = QFA_PX_F16 != 0;
    static constexpr int NP = F16 ? 2 : 3;                   // pieces per K-step
    static constexpr int KS_B = NP * 1024;
    static_assert(!F16 || KP <= 24, "the scales live in K slots 24..31 of K-step 0");
    static constexpr int S1_HALF = NKS * KS_B;               // bytes of one 16-pixel half: [K-step][piece][lane][8 k]
    static constexpr int TILE_B = 2 * S1_HALF;               // per 32-pixel tile
    static constexpr int NCHUNK = TILE_B / 1024;
};

// image of one 32-pixel tile per block: B[k = 32 ks + 8 g + j][px = 2 lo + h] as three bf16 pieces
template <int KP>
__global__ __launch_bounds__(256) void k_prep_px(const float *__restrict__ F, int Npix, int Nh,
                                                 unsigned char *__restrict__ PXI) {
    using X = PX<KP>;
    unsigned char *tile = PXI + (size_t)blockIdx.x * X::TILE_B;
    const int p0 = 32 * blockIdx.x;
    __shared__ float f[32][KP + 1];
    for (int i = threadIdx.x; i < 32 * KP; i += 256) {
        const int px = i / KP, a = i % KP;
        f[px][a] = (p0 + px < Npix && a < Nh) ? F[(size_t)(p0 + px) * Nh + a] : 0.f;
    }
    __syncthreads();
    __shared__ float tsc[32][3];                                  // F16: the pixel's power of two t, 1 / t, 1 / t^2
    if (X::F16 && threadIdx.x < 32) {
        float mx = 0.f;
        for (int a = 0; a < KP; ++a) mx = fmaxf(mx, fabsf(f[threadIdx.x][a]));
        int e = 7;
        if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);       // t f_a in [2^6, 2^7) for the largest: pairs below 2^14
        e = e < -50 ? -50 : (e > 60 ? 60 : e);
        tsc[threadIdx.x][0] = ldexpf(1.f, 7 - e);
        tsc[threadIdx.x][1] = ldexpf(1.f, e - 7);
        tsc[threadIdx.x][2] = ldexpf(1.f, 2 * (e - 7));
    }
    if (X::F16) __syncthreads();
    for (int i = threadIdx.x; i < 2 * X::NKS * 64; i += 256) {
        const int lane = i & 63, ks = (i >> 6) % X::NKS, h = i / (64 * X::NKS);
        const int lo = lane & 15, g = lane >> 4, px = 2 * lo + h;
        const float t1 = X::F16 ? tsc[px][0] : 1.f, t2 = t1 * t1;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * g + j;
            float x = 0.f;
            if (ks == 0) {
                if (kk < KP) x = f[px][kk] * t1;
            } else {
                const int q = 32 * (ks - 1) + kk;
                if (q < X::KK2) {
                    int a = 0;
                    while (a + 1 < KP && pair_index(a + 1, a + 1, KP) <= q) ++a;
                    const int b = a + (q - pair_index(a, a, KP));
                    x = f[px][a] * f[px][b] * t2;
                }
            }
            v[j] = x;
        }
        unsigned char *dst = tile + h * X::S1_HALF + ks * X::KS_B + lane * 16;
        if constexpr (X::F16) {
            u32x4 ph, pm;
            split8h(v, ph, pm);
            if (ks == 0 && g == 3) ph = u32x4{__float_as_uint(tsc[px][1]), __float_as_uint(tsc[px][2]), 0u, 0u};
            *reinterpret_cast<u32x4 *>(dst) = ph;
            *reinterpret_cast<u32x4 *>(dst + 1024) = pm;
        } else {
            u32x4 ph, pm, pl;
            split8(v, ph, pm, pl);
            *reinterpret_cast<u32x4 *>(dst) = ph;
            *reinterpret_cast<u32x4 *>(dst + 1024) = pm;
            *reinterpret_cast<u32x4 *>(dst + 2048) = pl;
        }
    }
}

#ifndef QFA_PX_REALIGN
#define QFA_PX_REALIGN 1    // N_h <= 8: stores of whole aligned lines when the rows of cont / unc do not start on one
#endif
struct __attribute__((packed, aligned(4))) pxf2 { float v[2]; };       // 4-byte aligned 8-byte store
#ifndef QFA_PX_NT
#define QFA_PX_NT 1         // cont / unc leave as non-temporal stores (never read back by this call): writer at c3 0.87 - 0.91 -> 0.82 ms, same box
#endif
typedef float pxv2 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void px_store2(float *dst, float a, float b) {
#if QFA_PX_NT
    __builtin_nontemporal_store(pxv2{a, b}, reinterpret_cast<pxv2 *>(dst));
#else
    *reinterpret_cast<pxf2 *>(dst) = pxf2{{a, b}};
#endif
}

// One work item = (block of 64 spectra, range of 32-pixel tiles); SOL as k_solve<KP, true> leaves it
// ([hmean | hcov' with doubled off-diagonals]).
#ifndef QFA_PX_ABL
#define QFA_PX_ABL 0        // timing-only ablations of the whole-tile path: 1 no stores, 2 no MFMAs, 4 no image DMA behind the first
#endif
#ifndef QFA_PX_OCC2
#define QFA_PX_OCC2 2       // workgroups per CU of the two-groups-per-wave form
#endif
#ifndef QFA_PX_SINGLE_B
#define QFA_PX_SINGLE_B 1
#endif
// host side of the re-aligned store path's condition (see k_predict_x): rows that do not start on a 64-byte boundary
inline bool px_realign(int KP, int Npix, const void *cont, const void *unc) {
    const size_t ca = reinterpret_cast<size_t>(cont), ua = reinterpret_cast<size_t>(unc);
    return KP == 8 && QFA_PX_REALIGN && ((ca ^ ua) & 127) == 0 && ((ca & 63) != 0 || (Npix & 15) != 0);
}
template <int KP, int SPW = 1, bool RA = false>   // SPW: groups of 16 spectra per wave (2: every B-operand read from LDS feeds two
                                                  // MFMA chains); RA: re-aligned stores (N_h <= 8; its own instantiation: 156
                                                  // registers against 92 would cost the aligned shapes their fourth workgroup per CU)
__global__ __launch_bounds__(256, (KP == 16 && SPW == 1) ? (QFA_PX_SINGLE_B ? 4 : 3) : ((KP == 8 && SPW == 1) ? 4 : QFA_PX_OCC2)) void k_predict_x(const float *__restrict__ mu, int B, int Npix, int ntiles,
                                                      WorkPlan wp, const unsigned char *__restrict__ PXI,
                                                      const float *__restrict__ SOL, float *__restrict__ cont,
                                                      float *__restrict__ unc) {
    using C = Cfg<KP>;
    using X = PX<KP>;
    // Shipped forms: KP = 16 with SPW = 2 (QFA_PX_SPW) -- a wave holds the operands of TWO groups of 16 spectra (232 VGPRs), a
    // workgroup = 128 spectra, two per CU, ring of two whole tiles (72 KiB), one barrier per tile, every B read from LDS feeds
    // two alternating MFMA chains (c3: 1.04 -> 0.90 ms); KP = 8 with SPW = 1: ring of two whole tiles (36 KiB), four per CU.
    // KP = 16 with SPW = 1 (the form before): the ring holds two HALVES of a tile (36 KiB instead of 72: four workgroups per
    // CU instead of two; 118 VGPRs allow it) and a tile step is two half-steps with a barrier each.
    constexpr bool HR = KP == 16 && SPW == 1;
    __shared__ __attribute__((aligned(16))) unsigned char lds[HR ? 2 * X::S1_HALF : 2 * X::TILE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = wave_uniform(tid >> 6);
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int n = t1 - t0;
    const int s0 = (blk * 4 + wv) * 16 * SPW;
    const bool active = s0 < B;
    const int lo = lane & 15, g = lane >> 4;
    // A operand: spectrum s0 + lo, k = 32 ks + 8 g + j  (SPW = 2: a second set for the spectra s0 + 16 + lo)
    constexpr bool F16 = X::F16;
    u32x4 S1h[X::NKS], S1m[X::NKS], S1l[F16 ? 1 : X::NKS];
    u32x4 T1h[SPW == 2 ? X::NKS : 1], T1m[SPW == 2 ? X::NKS : 1], T1l[(SPW == 2 && !F16) ? X::NKS : 1];
    // F16: the inverse powers of two of the spectra of the lane's OUTPUT rows (4 g + r): hmean, hcov'
    float ism[SPW][4], isq[SPW][4];
#pragma unroll
    for (int grp = 0; grp < SPW; ++grp) {
        const bool v = active && (s0 + 16 * grp + lo) < B;
        const float *sol = SOL + (size_t)(v ? s0 + 16 * grp + lo : 0) * C::NSOL;
        auto value = [&](int ks, int j) __attribute__((always_inline)) {
            const int kk = 8 * g + j;
            float val = 0.f;
            if (ks == 0) {
                if (v && kk < KP) val = sol[kk];
            } else {
                const int q = 32 * (ks - 1) + kk;
                if (v && q < X::KK2) val = sol[C::SOL_CI + q];
            }
            return val;
        };
        float xs[X::NKS][8];                              // (read once)
#pragma unroll
        for (int ks = 0; ks < X::NKS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) xs[ks][j] = value(ks, j);
        float scm = 1.f, scq = 1.f;
        if constexpr (F16) {
            float mm = 0.f, mq = 0.f;
#pragma unroll
            for (int ks = 0; ks < X::NKS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (ks == 0) mm = fmaxf(mm, fabsf(xs[ks][j])); else mq = fmaxf(mq, fabsf(xs[ks][j]));
                }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) { mm = fmaxf(mm, __shfl_xor(mm, o)); mq = fmaxf(mq, __shfl_xor(mq, o)); }
            float im, iq;
            scm = f16_row_scale(mm, im);
            scq = f16_row_scale(mq, iq);
#pragma unroll
            for (int r = 0; r < 4; ++r) { ism[grp][r] = __shfl(im, 4 * g + r); isq[grp][r] = __shfl(iq, 4 * g + r); }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) ism[grp][r] = isq[grp][r] = 1.f;
        }
#pragma unroll
        for (int ks = 0; ks < X::NKS; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = xs[ks][j] * (ks == 0 ? scm : scq);
            if constexpr (F16) {
                u32x4 a, b;
                split8h(x, a, b);
                if (grp == 0) { S1h[ks] = a; S1m[ks] = b; }
                else { T1h[SPW == 2 ? ks : 0] = a; T1m[SPW == 2 ? ks : 0] = b; }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned a, b, c;
                    split2(x[2 * q], x[2 * q + 1], a, b, c);
                    if (grp == 0) { S1h[ks][q] = a; S1m[ks][q] = b; S1l[F16 ? 0 : ks][q] = c; }
                    else { T1h[SPW == 2 ? ks : 0][q] = a; T1m[SPW == 2 ? ks : 0][q] = b; T1l[(SPW == 2 && !F16) ? ks : 0][q] = c; }
                }
            }
        }
    }
    // F16: the lane's pixel of half h of the tile in ring slot `img`: 1 / t, 1 / t^2 (k_prep_px)
    auto pixel_scales = [&](const unsigned char *half) __attribute__((always_inline)) {
        typedef float f32x2t __attribute__((ext_vector_type(2)));
        return F16 ? *reinterpret_cast<const f32x2t *>(half + (48 + lo) * 16) : f32x2t{1.f, 1.f};
    };
    const bool full_wave = active && s0 + 16 * SPW <= B;
    // mu of the lane's two pixels of tile tg, requested one tile ahead by asm loads IN FRONT of the image DMA of that tile:
    // the counted wait that retires the DMA retires them (as ordinary loads in the loop they made hipcc wait vmcnt(0) in
    // the middle of the step -- for the DMA just issued and the previous tile's stores)
    float mn0 = 0.f, mn1 = 0.f;
    auto load_mu = [&](int tg) {
        const int p0 = min(32 * tg + 2 * lo, Npix - 1), p1 = min(32 * tg + 2 * lo + 1, Npix - 1);
        if (QFA_TRACKED_LOADS) {
            mn0 = mu[p0];
            mn1 = mu[p1];
        } else {
            aload4(mn0, mu, 4u * (unsigned)p0);
            aload4(mn1, mu, 4u * (unsigned)p1);
        }
    };
    auto land_mu = [&]() { asm volatile("" : "+v"(mn0), "+v"(mn1)); };
    // NP 1-KiB pieces of an image by the four waves: each wave a CONTIGUOUS run behind one write of M0 (waves below NP % 4 one
    // piece more: (NP - wv + 3) / 4 pieces, the count the waits below assume).  Piece by piece with its own M0 every request
    // waited until the texture path had accepted the one before (glds16_run, qfa_xdl_kernels.h), and a piece count that is
    // no multiple of 4 cost a scalar branch per piece.
    auto move_pieces = [&](auto np_tag, const unsigned char *sbase, const unsigned char *dst) {
        constexpr int NP = decltype(np_tag)::value, LO = NP / 4, EX = NP % 4;
        static_assert(LO >= 1, "at least one piece per wave");
        const int first = wv * LO + min(wv, EX);
        const unsigned char *src = sbase + first * 1024;
        const unsigned d = wave_uniform(lds_addr(dst + first * 1024));
        if (EX && wv < EX) glds16_runs<LO + 1>(src, (unsigned)lane * 16u, d);
        else glds16_runs<LO>(src, (unsigned)lane * 16u, d);
    };
    auto get_tile = [&](int c) {
        const unsigned char *src = PXI + (size_t)(t0 + c) * X::TILE_B;
        const unsigned long long a = reinterpret_cast<unsigned long long>(src);
        const unsigned char *sbase = reinterpret_cast<const unsigned char *>(
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
        move_pieces(std::integral_constant<int, X::NCHUNK>{}, sbase, lds + (c & 1) * X::TILE_B);
    };
    if constexpr (HR) {
        if (n <= 0) return;
        constexpr int NCH = X::S1_HALF / 1024;
        auto get_half = [&](int u) {
            const unsigned char *sbase = uniform_ptr(PXI + (size_t)(t0 + (u >> 1)) * X::TILE_B + (u & 1) * X::S1_HALF);
            move_pieces(std::integral_constant<int, NCH>{}, sbase, lds + (u & 1) * X::S1_HALF);
        };
        // Schedule of the requests (round 3).  The eight stores of a tile need an HBM write round trip; in the round-2
        // order the image DMA of the next half-step was issued BEHIND them, and since vmcnt retires in issue order the wait
        // for that DMA in front of the next barrier was a wait for the stores -- one write latency per tile on the
        // critical path of every workgroup.  Now everything a tile step requests is issued IN FRONT of its stores:
        //   half-step (c, 1):  mu(c + 1), image (c + 1, 0) -> slot 0 | MFMAs on slot 1 | barrier | image (c + 1, 1) -> slot 1 |
        //                      stores of tile c | wait: all but [image (c + 1, 1), stores(c)] | barrier
        //   half-step (c + 1, 0):  MFMAs on slot 0 | wait: all but stores(c) | barrier
        // so the stores of tile c are first waited for at the end of half-step (c + 1, 1): three half-steps later.  Price: a
        // third barrier per tile (slot 1 must be free before it is refilled in the same half-step).
        const int n_p = (NCH - wv + 3) / 4;                  // image pieces of one half moved by this wave
        load_mu(t0);
        get_half(0);
        get_half(1);
        dma_wait<0>();
        wg_barrier();
        asm volatile("" ::: "memory");
        float co0[4], un0[4];
        bool prev_counted = false;                           // the previous tile's stores are the 8 youngest requests
        auto stage1 = [&](int slot, f32x4 &afy, f32x4 &aq) {
            const unsigned char *bp = lds + slot * X::S1_HALF + lane * 16;
            afy = f32x4{0.f, 0.f, 0.f, 0.f};
            aq = f32x4{0.f, 0.f, 0.f, 0.f};
            const auto its = pixel_scales(lds + slot * X::S1_HALF);
#pragma unroll
            for (int ks = 0; ks < X::NKS; ++ks) {
                const u32x4 bh = *reinterpret_cast<const u32x4 *>(bp + ks * X::KS_B),
                            bm = *reinterpret_cast<const u32x4 *>(bp + ks * X::KS_B + 1024);
                if constexpr (F16) {
                    if (ks == 0) afy = xdl3h(S1h[ks], S1m[ks], bh, bm, afy);
                    else aq = xdl3h(S1h[ks], S1m[ks], bh, bm, aq);
                } else {
                    const u32x4 bl = *reinterpret_cast<const u32x4 *>(bp + ks * X::KS_B + 2048);
                    if (ks == 0) afy = xdl6(S1h[ks], S1m[ks], S1l[F16 ? 0 : ks], bh, bm, bl, afy);
                    else aq = xdl6(S1h[ks], S1m[ks], S1l[F16 ? 0 : ks], bh, bm, bl, aq);
                }
                __builtin_amdgcn_sched_barrier(0);           // (four waves per SIMD hide the LDS latency: one B buffer)
            }
            if constexpr (F16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    afy[r] = (afy[r] * ism[0][r]) * its[0];
                    aq[r] = (aq[r] * isq[0][r]) * its[1];
                }
            }
        };
        for (int c = 0; c < n; ++c) {
            const int tg = t0 + c;
            const bool more = c + 1 < n;
            land_mu();
            const float mc[2] = {mn0, mn1};
            // ---- half-step (c, 0)
            if (active) {
                f32x4 afy, aq;
                stage1(0, afy, aq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    co0[r] = afy[r] + mc[0];
                    un0[r] = __builtin_amdgcn_sqrtf(aq[r]);
                }
            }
            if (prev_counted) dma_wait<8>();                 // image (c, 1) has landed; the stores of tile c - 1 may still fly
            else dma_wait<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wg_barrier();
            asm volatile("" ::: "memory");
            // ---- half-step (c, 1)
            if (more) {
                load_mu(tg + 1);
                get_half(2 * c + 2);                         // slot 0 is free behind the barrier
            }
            f32x4 afy1 = {0.f, 0.f, 0.f, 0.f}, aq1 = {0.f, 0.f, 0.f, 0.f};
            if (active) stage1(1, afy1, aq1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wg_barrier();                                    // every wave has read slot 1
            asm volatile("" ::: "memory");
            int q1 = 0;
            if (more) {
                get_half(2 * c + 3);
                q1 = n_p;
            }
            bool counted = false;
            if (active) {
                const float m = mc[1];
                const int px = 32 * tg + 2 * lo;
                if (full_wave && 32 * tg + 31 < Npix) {      // wave-uniform: exactly eight store instructions
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t o = (size_t)(s0 + 4 * g + r) * Npix + px;
                        px_store2(cont + o, co0[r], afy1[r] + m);
                        px_store2(unc + o, un0[r], __builtin_amdgcn_sqrtf(aq1[r]));
                    }
                    counted = true;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = s0 + 4 * g + r;
                        if (s < B && px < Npix) {
                            cont[(size_t)s * Npix + px] = co0[r];
                            unc[(size_t)s * Npix + px] = un0[r];
                        }
                        if (s < B && px + 1 < Npix) {
                            cont[(size_t)s * Npix + px + 1] = afy1[r] + m;
                            unc[(size_t)s * Npix + px + 1] = __builtin_amdgcn_sqrtf(aq1[r]);
                        }
                    }
                }
            }
            // image (c + 1, 0) and mu(c + 1) were issued in front of the image (c + 1, 1) pieces and the stores
            if (counted) dma_wait_n(8 + q1);
            else dma_wait<0>();
            prev_counted = counted;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wg_barrier();
            asm volatile("" ::: "memory");
        }
        return;
    }
    if (n > 0) {
        load_mu(t0);
        get_tile(0);
    }
    dma_wait<0>();
    wg_barrier();
    asm volatile("" ::: "memory");
    // ---- re-aligned stores (N_h <= 8, one group per wave).  A tile's 32 pixels of a row are 128 bytes that straddle two lines
    // unless the row starts on a line: with N_pix = 1913 or 9243 (the reference's models) every store instruction writes eight
    // partial lines, and the writer runs at 2.8-3.4 TB/s where rows of 1920 or 2000 pixels reach 4.1-4.2 (tools/writer_align.py).
    // Here step t stores, per row, the ALIGNED line that contains the tile's first pixel: pixels 32 t - phi + x, x = 0..31, with
    // phi = the row's offset into its line in pixels -- the last phi pixels of tile t - 1 (kept in registers) and the first
    // 32 - phi of tile t.  Lane lo writes x = 2 lo, 2 lo + 1: the values come from the row group's other lanes by two
    // ds_bpermute (the SOURCE lane picks current or previous tile and which of its two pixels serves a first / second dword:
    // every value of a lane is consumed exactly once).  One more step behind the last tile flushes the tails.
    static_assert(!RA || (KP == 8 && SPW == 1), "re-aligned stores: N_h <= 8, one group per wave");
    const size_t ca = reinterpret_cast<size_t>(cont);
    constexpr bool realign = RA;                                  // (the host picks the instantiation: px_realign)
    float cop[2][4], unp[2][4];                                   // the previous tile's values
#pragma unroll
    for (int r = 0; r < 4; ++r) cop[0][r] = cop[1][r] = unp[0][r] = unp[1][r] = 0.f;
    // the line offset of row s0 + 4 g + r in pixels is (phi0 + r N_pix) & 31 (recomputed per use: registers)
    const int phi0 = (int)(((ca >> 2) + (size_t)(s0 + 4 * g) * (size_t)Npix) & 31);
    const int Pbeg = 32 * t0, Pend = min(32 * t1, Npix);
    // returns true when exactly eight 8-byte stores were issued (wave-uniform)
    auto realign_store = [&](int t, bool have_cur, const float (*co)[4], const float (*un)[4]) {
        const bool fast = have_cur && full_wave && t > t0 && 32 * t + 32 <= Pend;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ph = (phi0 + r * Npix) & 31, s = s0 + 4 * g + r;
            const bool odd = ph & 1, c0now = 2 * lo + ph <= 31, c1now = 2 * lo + 1 + ph <= 31;
            const int L0 = ((2 * lo - ph) & 31) >> 1, L1 = ((2 * lo + 1 - ph) & 31) >> 1;
            const int P0 = 32 * t - ph + 2 * lo;
#pragma unroll
            for (int arr = 0; arr < 2; ++arr) {
                const float cur0 = arr ? un[0][r] : co[0][r], cur1 = arr ? un[1][r] : co[1][r];
                const float prv0 = arr ? unp[0][r] : cop[0][r], prv1 = arr ? unp[1][r] : cop[1][r];
                const float send0 = c0now ? cur0 : prv0, send1 = c1now ? cur1 : prv1;
                const float slot0 = odd ? send1 : send0, slot1 = odd ? send0 : send1;
                const float d0 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (16 * g + L0), __float_as_int(slot0)));
                const float d1 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (16 * g + L1), __float_as_int(slot1)));
                float *q = (arr ? unc : cont) + ((size_t)s * (size_t)Npix + (size_t)(long)P0);
                if (fast) px_store2(q, d0, d1);
                else {
                    if (s < B && P0 >= Pbeg && P0 < Pend) q[0] = d0;
                    if (s < B && P0 + 1 >= Pbeg && P0 + 1 < Pend) q[1] = d1;
                }
            }
        }
        if (have_cur) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cop[0][r] = co[0][r]; cop[1][r] = co[1][r];
                unp[0][r] = un[0][r]; unp[1][r] = un[1][r];
            }
        }
        return fast;
    };
    for (int c = 0; c < n; ++c) {
        land_mu();
        const float mc[2] = {mn0, mn1};
        if (c + 1 < n) {
            load_mu(t0 + c + 1);
            if (!(QFA_PX_ABL & 4)) get_tile(c + 1);
        }
        bool counted = false;
        if (active) {
            const int tg = t0 + c;
            const unsigned char *img = lds + (c & 1) * X::TILE_B;
            float co[2][4], un[2][4], co2[2][4], un2[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char *bp = img + h * X::S1_HALF + lane * 16;
                f32x4 afy = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f}, afy2 = {0.f, 0.f, 0.f, 0.f}, aq2 = {0.f, 0.f, 0.f, 0.f};
                constexpr int NP = X::NP;
                u32x4 bq[2][NP];
                const auto its = pixel_scales(img + h * X::S1_HALF);
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) bq[0][pc] = *reinterpret_cast<const u32x4 *>(bp + pc * 1024);
#pragma unroll
                for (int ks = 0; ks < X::NKS; ++ks) {
                    if (ks + 1 < X::NKS) {
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            bq[(ks + 1) & 1][pc] = *reinterpret_cast<const u32x4 *>(bp + (ks + 1) * X::KS_B + pc * 1024);
                    }
                    const u32x4 &bh = bq[ks & 1][0], &bm = bq[ks & 1][1], &bl = bq[ks & 1][NP - 1];
                    if (QFA_PX_ABL & 2) {
                        afy[0] += __uint_as_float(bh[0] ^ bm[1] ^ bl[2]);
                    } else if constexpr (F16) {
                        f32x4 &c0 = ks == 0 ? afy : aq, &c1 = ks == 0 ? afy2 : aq2;
                        if constexpr (SPW == 2) {       // two chains alternating (xdl3h's order)
                            c0 = xdlh(S1h[ks], bm, c0); c1 = xdlh(T1h[ks], bm, c1);
                            c0 = xdlh(S1m[ks], bh, c0); c1 = xdlh(T1m[ks], bh, c1);
                            c0 = xdlh(S1h[ks], bh, c0); c1 = xdlh(T1h[ks], bh, c1);
                        } else c0 = xdl3h(S1h[ks], S1m[ks], bh, bm, c0);
                    } else if constexpr (SPW == 2) {    // two chains alternating (six_terms' order)
                        f32x4 &c0 = ks == 0 ? afy : aq, &c1 = ks == 0 ? afy2 : aq2;
                        constexpr int kl = F16 ? 0 : ks;
                        c0 = xdl(S1h[ks], bl, c0); c1 = xdl(T1h[ks], bl, c1);
                        c0 = xdl(S1l[kl], bh, c0); c1 = xdl(T1l[kl], bh, c1);
                        c0 = xdl(S1m[ks], bm, c0); c1 = xdl(T1m[ks], bm, c1);
                        c0 = xdl(S1m[ks], bh, c0); c1 = xdl(T1m[ks], bh, c1);
                        c0 = xdl(S1h[ks], bm, c0); c1 = xdl(T1h[ks], bm, c1);
                        c0 = xdl(S1h[ks], bh, c0); c1 = xdl(T1h[ks], bh, c1);
                    } else if (ks == 0) afy = xdl6(S1h[ks], S1m[ks], S1l[F16 ? 0 : ks], bh, bm, bl, afy);
                    else aq = xdl6(S1h[ks], S1m[ks], S1l[F16 ? 0 : ks], bh, bm, bl, aq);
                }
                if constexpr (F16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        afy[r] = (afy[r] * ism[0][r]) * its[0];
                        aq[r] = (aq[r] * isq[0][r]) * its[1];
                        if (SPW == 2) {
                            afy2[r] = (afy2[r] * ism[SPW - 1][r]) * its[0];
                            aq2[r] = (aq2[r] * isq[SPW - 1][r]) * its[1];
                        }
                    }
                }
                const float m = mc[h];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    co[h][r] = afy[r] + m;
                    un[h][r] = __builtin_amdgcn_sqrtf(aq[r]);
                    if (SPW == 2) {
                        co2[h][r] = afy2[r] + m;
                        un2[h][r] = __builtin_amdgcn_sqrtf(aq2[r]);
                    }
                }
            }
            const int px = 32 * tg + 2 * lo;
            if (RA && realign) {
                counted = realign_store(tg, true, co, un);
            } else if (QFA_PX_ABL & 1) {
                float acc_ = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_ += co[0][r] + co[1][r] + un[0][r] + un[1][r] + (SPW == 2 ? co2[0][r] + co2[1][r] + un2[0][r] + un2[1][r] : 0.f);
                if (acc_ == 1.2345e-30f) cont[0] = acc_;
            } else if (full_wave && 32 * tg + 31 < Npix) {          // wave-uniform: exactly 8 SPW store instructions
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t o = (size_t)(s0 + 4 * g + r) * Npix + px;
                    px_store2(cont + o, co[0][r], co[1][r]);
                    px_store2(unc + o, un[0][r], un[1][r]);
                }
                if (SPW == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t o = (size_t)(s0 + 16 + 4 * g + r) * Npix + px;
                        px_store2(cont + o, co2[0][r], co2[1][r]);
                        px_store2(unc + o, un2[0][r], un2[1][r]);
                    }
                }
                counted = true;
            } else {
#pragma unroll
                for (int grp = 0; grp < SPW; ++grp) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int s = s0 + 16 * grp + 4 * g + r;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (s < B && px + h < Npix) {
                                cont[(size_t)s * Npix + px + h] = grp ? co2[h][r] : co[h][r];
                                unc[(size_t)s * Npix + px + h] = grp ? un2[h][r] : un[h][r];
                            }
                        }
                    }
                }
            }
        }
        // the image pieces of tile c + 1 were issued before this step's stores: all but the 8 SPW stores must be done
        if (counted) dma_wait<8 * SPW>();
        else dma_wait<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();
        asm volatile("" ::: "memory");
    }
    if (RA && realign && active && n > 0) realign_store(t1, false, cop, unp);      // the tails of the last tile's rows
}
