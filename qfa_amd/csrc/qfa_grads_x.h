// qfa_grads_x.h -- pass 2 (gradients) for N_h = 9..16 with EVERY contraction on the bf16 XDL pipe (gfx950).
//
// Why a second form of pass 2 (k_grads of qfa_step_kernels.h stays for the other widths):
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
// the per-pixel arithmetic can hold at two waves per SIMD.  So the work of a group is split between two waves:
//   role A  stage 1 + stage 2 of tile c   : holds [y | Cinv'] pieces, streams the spectra, VALU-heavy
//   role B  stage 3 of tile c - 1         : holds the Z / p pieces, XDL-heavy; also the image DMA and the flushes
// and beta / gamma travel A -> B through LDS.
//
// Workgroup = 512 threads = 4 role-A waves + 4 role-B waves = 64 spectra, one workgroup per CU: waves w and w + 4 (one
// group of 16 spectra, roles A and B) share a SIMD, so the VALU-heavy and the XDL-heavy instruction streams overlap on
// it by themselves.  (Two 256-thread workgroups per CU were tried: the co-resident workgroups' waves land on the SIMDs
// in the same role order, VALU beside VALU and XDL beside XDL: 4.25 ms against 3.5.)
// One workgroup per CU in lockstep has nothing to run while it waits for memory, and registers for one tile of
// prefetch exposed the whole HBM latency (profiles/r2_ablation_k_grads_x.txt).  So the spectra do not go through
// registers: every role-A wave streams the delta / sigma / zabs rows of its 16 spectra into its own LDS staging
// buffers by LDS-DMA, TWO tiles ahead (2 x 6 KiB per wave), and copies the tile it is about to process into registers
// at the start of the step (the masks, staged the same way, are folded into the sign of sigma there).
// The LDS for that comes from the image ring, which holds HALF tiles (16 pixels, 19 KiB): role A consumes half h of tile
// c in half-step t = 2 c + h from ring slot h, role B issues the DMA of half t + 1 at the start of half-step t; one
// barrier per half-step.
//
// Tile = 32 pixels.  Pixel index inside a tile: role A's lane (lo = lane & 15, g = lane >> 4) owns pixels 2 lo + h
// (h = 0, 1: one 8-byte load per array and spectrum) of the spectra 4 g + r; half h of the stage-1 image holds the
// pixels 2 lo + h in column lo.
//
// Memory protocol (the one k_moments_x uses): every DMA is an asm statement, invisible to hipcc's s_waitcnt
// bookkeeping -- with a tracked LDS-DMA in flight hipcc waits vmcnt(0) at the next use of ANY load result and again at
// __syncthreads().  Role A's queue holds, per tile, its 8 staging DMAs (issued in step c for tile c + 2; waited for at
// the start of step c + 2 with a counted vmcnt that leaves tile c + 3's in flight); role B's holds the image DMA pieces
// and behind them the flush atomics, which a counted vmcnt in front of the barrier leaves in flight for one more
// half-step.  Nothing on the hot path loads into registers through asm (an earlier form did, and hipcc moved copies
// of such registers in front of the wait; tools/audit_asm_loads.py still checks the compiled code for that), and
// `make check-gx` fails on any scratch use (spilled loop-carried registers came
// back wrong from run to run in an earlier build -- tools/diag_slab.py).
#pragma once
#include "qfa_common.h"
#include "qfa_xdl_kernels.h"      // glds16a / dma_wait / lds_addr: the untracked LDS-DMA and counted-vmcnt helpers

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef QFA_GX_F16
#define QFA_GX_F16 1
#endif
template <int KP_>
struct GXT {                                             // KP = 16 (N_h = 9..16) or 8 (N_h <= 8)
    static constexpr int KP = KP_, KK2 = KP * (KP + 1) / 2;
    static constexpr int NKS = 1 + (KK2 + 31) / 32;      // K-steps of stage 1: [y, 0 | pair products, 32 per step]: 6 / 3
    // Round 5 (QFA_GX_F16): stage 1 on TWO float16 pieces per operand, three products per K-step (qfa_common.h "float16 pieces");
    // the image holds t f_a and t^2 f_a f_b, 1 / t and 1 / t^2 of a half's pixels in its parameter KiB (floats 80.., 96..)
    static constexpr bool F16 = QFA_GX_F16 != 0;
    static constexpr int NP = F16 ? 2 : 3;               // pieces per K-step
    static constexpr int KS_B = NP * 1024;
    static constexpr int PAR_IT1 = 80, PAR_IT2 = 96;     // float index in a half's parameter KiB: 1 / t, 1 / t^2 of pixel 2 lo + h
    static constexpr int S1_HALF = NKS * KS_B;           // bytes of the stage-1 image of one 16-pixel half
    static constexpr int HALF_B = S1_HALF + 1024;        // ring slot: + float32 Psi[16], omega[16] of its pixels (19 / 10 KiB)
    static constexpr int OFF_FP = 2 * HALF_B;            // F as bf16 pieces, A operand of stage 3: [piece][lane][8 a]
    static constexpr int TILE_B = OFF_FP + 3 * 1024;     // 41 / 23 KiB per 32-pixel tile in global memory
    static constexpr int NCH_HALF = HALF_B / 1024;       // one-KiB DMA pieces per half (+ 3 for the F pieces with h = 1)
    static constexpr int GROW = 16;                      // floats per row of the transposed gamma slot
    static constexpr int FROW = KP + 4;                  // W form: floats per pixel row of the F block (conflict-free b128 reads)
    static constexpr int NG = 4;                         // groups of 16 spectra per workgroup
    static constexpr int SPB = 16 * NG;                  // spectra per workgroup
    static constexpr int STG_ARR = 16 * 128;             // staging: one array of one tile, [16 rows][32 px] float
    static constexpr int STG_MASK = 3 * STG_ARR;         // mask bytes [16 rows][32 px]
    static constexpr int STG_B = 3 * STG_ARR + 512;      // delta | sigma | zabs | mask (6.5 KiB per wave and tile)
    static constexpr int PARTF = 32 * KP;                // floats of one group's stage-3 sums of a tile: [32 px][KP b]
    // LDS (bytes), per workgroup
    static constexpr int L_IMG = 0;                                  // [2 halves][HALF_B]
    static constexpr int L_FP = L_IMG + 2 * HALF_B;                  // [2 tile parity][3 KiB]
    static constexpr int L_BETA = L_FP + 2 * 3072;                   // [2 tile parity][NG][16 s][32 px] float
    static constexpr int L_GAM = L_BETA + 2 * NG * 2048;             // [2][NG][32 rows][GROW] float
    static constexpr int L_PART = L_GAM + 2 * NG * 32 * GROW * 4;    // [2][NG][32 px][KP b] float
    static constexpr int L_PSUM = L_PART + 2 * NG * PARTF * 4;       // [2][NG][4 sums][32 px] float (summed over the wave)
    static constexpr int L_SCAL = L_PSUM + 2 * NG * 512;             // [NG waves][3 sums][64 lanes] double (role A)
    static constexpr int L_STG = L_SCAL + NG * 3 * 64 * 8;           // [NG waves][2 tile parity][STG_B]
    static constexpr int L_ZS = L_STG + NG * 2 * STG_B;              // [NG waves][16 spectra] float4: factored-z per-spectrum factors
    static constexpr int L_ROWS = L_ZS + NG * 256;                   // [NG waves][16 spectra] unsigned: rows of the batch arrays (ABI v3)
    static constexpr int L_TOTAL = L_ROWS + NG * 64;
};
static_assert(GXT<16>::L_TOTAL <= 160 * 1024 && GXT<8>::L_TOTAL <= 160 * 1024, "k_grads_x LDS");
static_assert(32 * GXT<16>::FROW * 4 <= 3072, "F block of the W form fits the F slot");
__device__ __forceinline__ f32x16 xdl32(const u32x4 &a, const u32x4 &b, f32x16 c) {     // 32x32x16
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                   0);
}
// TERMS: bf16 piece products per stage-3 contraction (qfa_common.h): 6 (float32-grade, the default) or 3 (QFA_F_S3_FAST)
template <int TERMS>
__device__ __forceinline__ f32x16 xdl32_6(const u32x4 &ah, const u32x4 &am, const u32x4 &al, const u32x4 &bh,
                                          const u32x4 &bm, const u32x4 &bl, f32x16 c) {
    c = xdl32(ah, bm, c);           // (first: its operands stay live -- see six_terms in qfa_common.h)
    if (TERMS == 6) {
        c = xdl32(al, bh, c);
        c = xdl32(ah, bl, c);
    }
    if (TERMS >= 4) c = xdl32(am, bm, c);
    c = xdl32(am, bh, c);
    return xdl32(ah, bh, c);
}

// ------------------------------------------------------------------------------------------------
// k_prep_pgx : F, Psi, omega -> the pass-2 image, one block per 32-pixel tile.
//   half h (h = 0, 1)  [K-step ks][piece][lane (g, lo)][8 k] bf16: B[k = 32 ks + 8 g + j][px = 2 lo + h]
//                      ks = 0: k < KP -> F[px][k]; ks >= 1: pair q = 32 (ks - 1) + 8 g + j -> F[px][a_q] F[px][b_q]
//                      then float32 Psi[lo], omega[lo] of the pixels 2 lo + h
//   stage-3 part       [piece][lane (r, h2)][8 a] bf16: A[px = r][a = 8 h2 + j]
// ------------------------------------------------------------------------------------------------
template <int KP>
__device__ __forceinline__ void prep_pgx_body(int bid, const float *__restrict__ F, const float *__restrict__ Psi,
                                              const float *__restrict__ omega, const ZPSrc &ZP, int Npix, int Nb, int Nh,
                                              int wform, unsigned char *__restrict__ PGX) {
    using GX = GXT<KP>;
    unsigned char *tile = PGX + (size_t)bid * GX::TILE_B;
    const int p0 = 32 * bid;
    __shared__ float f[32][17];
    for (int i = threadIdx.x; i < 32 * 16; i += 256) {
        const int px = i >> 4, a = i & 15;
        f[px][a] = (p0 + px < Npix && a < Nh) ? F[(size_t)(p0 + px) * Nh + a] : 0.f;
    }
    __syncthreads();
    __shared__ float tsc[32][3];                                  // F16: the pixel's power of two t, 1 / t, 1 / t^2
    if (GX::F16 && threadIdx.x < 32) {
        float mx = 0.f;
        for (int a = 0; a < KP; ++a) mx = fmaxf(mx, fabsf(f[threadIdx.x][a]));
        int e = 7;
        if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);       // t f_a in [2^6, 2^7) for the largest: pairs below 2^14
        e = e < -50 ? -50 : (e > 60 ? 60 : e);
        tsc[threadIdx.x][0] = ldexpf(1.f, 7 - e);
        tsc[threadIdx.x][1] = ldexpf(1.f, e - 7);
        tsc[threadIdx.x][2] = ldexpf(1.f, 2 * (e - 7));
    }
    if (GX::F16) __syncthreads();
    for (int i = threadIdx.x; i < 2 * GX::NKS * 64; i += 256) {
        const int lane = i & 63, ks = (i >> 6) % GX::NKS, h = i / (64 * GX::NKS);
        const int lo = lane & 15, g = lane >> 4, px = 2 * lo + h;
        const float t1 = GX::F16 ? tsc[px][0] : 1.f, t2 = t1 * t1;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * g + j;
            float x = 0.f;
            if (ks == 0) {
                if (kk < KP) x = f[px][kk] * t1;
            } else {
                const int q = 32 * (ks - 1) + kk;
                if (q < GX::KK2) {
                    int a = 0;
                    while (a + 1 < KP && pair_index(a + 1, a + 1, KP) <= q) ++a;
                    const int b = a + (q - pair_index(a, a, KP));
                    x = f[px][a] * f[px][b] * t2;
                }
            }
            v[j] = x;
        }
        unsigned char *dst = tile + h * GX::HALF_B + ks * GX::KS_B + lane * 16;
        if constexpr (GX::F16) {
            u32x4 ph, pm;
            split8h(v, ph, pm);
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
    // Psi, omega of each half's 16 pixels (+ zero padding of the KiB)
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int h = i >> 8, j = i & 255;
        float *po = reinterpret_cast<float *>(tile + h * GX::HALF_B + GX::S1_HALF);
        const int px = p0 + 2 * (j & 15) + h;
        float v = 0.f;
        if (j < 16) v = px < Npix ? Psi[px] : 0.f;
        else if (j < 32) v = px < Nb ? omega[px] : 0.f;
        else if (j < 80 && ZP.on() && px < Nb) {            // factored-z form: ti | pwi | l2i of the half's pixels
            const float4 q = ZP.at(px);
            v = j < 48 ? q.x : (j < 64 ? q.y : q.z);
        } else if (GX::F16 && j >= GX::PAR_IT1 && j < GX::PAR_IT2 + 16) v = tsc[2 * (j & 15) + h][j < GX::PAR_IT2 ? 1 : 2];
        po[j] = v;
    }
    if (wform) {       // stage 3 in its W form (role B, TERMS = 6): F of the tile as float32 rows [pixel 0..31][FROW]
        float *fr = reinterpret_cast<float *>(tile + GX::OFF_FP);
        for (int i = threadIdx.x; i < 32 * GX::FROW; i += 256) {
            const int px = i / GX::FROW, a = i % GX::FROW;
            fr[i] = a < KP ? f[px][a] : 0.f;
        }
    } else if (threadIdx.x < 64) {
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
template <int KP>
__global__ __launch_bounds__(256) void k_prep_pgx(const float *__restrict__ F, const float *__restrict__ Psi,
                                                  const float *__restrict__ omega, const float4 *__restrict__ ZP,
                                                  int Npix, int Nb, int Nh, int wform, unsigned char *__restrict__ PGX) {
    prep_pgx_body<KP>(blockIdx.x, F, Psi, omega, zp_table(ZP), Npix, Nb, Nh, wform, PGX);
}

struct SpecA {                       // role A: one lane's 4 spectra x 2 pixels of a tile (sigma < 0: masked pixel)
    f32x2 d[4], sg[4], z[4];
};
// ------------------------------------------------------------------------------------------------
// k_grads_x.  One work item = (block of 64 spectra, range of 32-pixel tiles) (WorkPlan counted in such blocks).
// slab != NULL: deterministic mode -- the block's tile partials go to row blk of the slab (plain stores), its scalar
// sums to slabS[item][wave][3]; k_reduce_slab (qfa_step_kernels.h) adds the rows to accum in block order.
// ------------------------------------------------------------------------------------------------
#ifndef QFA_GX_STAMPS
#define QFA_GX_STAMPS 0    // diagnostic build (tools/gx_stamps.sh): s_memtime shares of one workgroup's tile steps
#endif
#if QFA_GX_STAMPS
__device__ unsigned long long qfa_gx_stamps[2 * 32];
#define GXS(i)                                                                                 \
    {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        st_[i] += (unsigned)(t_ - st_last);                                                    \
        st_last = t_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
#else
#define GXS(i) {}
#endif
template <int KP, bool HASA, int TERMS, bool ZF>
__global__ __launch_bounds__(512, 2) void k_grads_x(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau, int B, int Npix, int Nb,
                                                    int Nh, int ntiles, WorkPlan wp,
                                                    const unsigned char *__restrict__ PGX,
                                                    const float *__restrict__ SOL, const float4 *__restrict__ ZS,
                                                    float *__restrict__ accum,
                                                    float *__restrict__ slab, double *__restrict__ slabS,
                                                    int slab_stride, Scal64 *__restrict__ sc64) {
    using C = Cfg<KP>;
    using GX = GXT<KP>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[GX::L_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv8 = wave_uniform(tid >> 6);               // 0..7
    const bool roleA = wv8 < GX::NG;
    const int w = wv8 & (GX::NG - 1);                      // group of 16 spectra inside the block
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int s0 = blk * GX::SPB + w * 16;
    const bool active = s0 < B;                            // wave-uniform
    const int n = t1 - t0;
    const int nbt = (Nb + 31) >> 5;                        // tiles that contain blue pixels
    const DevConsts k = load_consts(p, tau);

#if QFA_GX_STAMPS
    unsigned st_[32];
    for (int i = 0; i < 32; ++i) st_[i] = 0;
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    const bool det = slab != nullptr;
    float *accF = det ? slab + (size_t)blk * (size_t)slab_stride : accum;
    float *accA = accF + (size_t)Npix * Nh;                // sumA | gPsi | gOmega | cnt (contiguous)
    float *accS = accum + (size_t)Npix * Nh + 3 * (size_t)Npix + Nb;

    // zero the slots that inactive groups never write
    for (int i = tid; i < (GX::L_SCAL - GX::L_BETA) / 4; i += 512) reinterpret_cast<float *>(lds + GX::L_BETA)[i] = 0.f;

    // de-phase the tile order between workgroups (concurrent flushes then hit different rows; the workgroups running
    // together still share a window of the image in L2)
    const int rot = n > 0 ? (int)(((unsigned)blk * 2654435761u) % (unsigned)min(n, 32)) : 0;
    auto tile_of = [&](int c) {
        int x = c + rot;
        if (x >= n) x -= n;
        return t0 + x;
    };

#ifndef QFA_GX_ABL
#define QFA_GX_ABL 0       // timing-only ablations (wrong results): 1 no spectra staging, 2 no flush, 4 no image DMA,
#endif                     // 8 the staging re-reads the first tile (cache hits), 16 zabs staged from the delta rows (16-byte aligned),
                           // 32 flush without its atomics, 64 flush without its LDS reads
#ifndef QFA_GX_BPRIO
#define QFA_GX_BPRIO 1        // priority of the role-B waves (0..3; 4 = 1 on red tiles only): 1 or 2 measured -0.05..-0.1 ms at c3
#endif
#ifndef QFA_GX_ROLE
#define QFA_GX_ROLE 0      // register-pressure experiments: 1 = role A only, 2 = role B only
#endif
    if (roleA && QFA_GX_ROLE != 2) {
        // ================================================================ role A: stage 1 + stage 2
        const int lo = lane & 15, g = lane >> 4;
        // A operand of stage 1: spectrum s0 + lo, k = 32 ks + 8 g + j
        u32x4 S1h[GX::NKS], S1m[GX::NKS], S1l[GX::F16 ? 1 : GX::NKS];
        float is0[4] = {1.f, 1.f, 1.f, 1.f}, is1[4] = {1.f, 1.f, 1.f, 1.f};      // F16: inverse powers of two of the spectra 4 g + r (y | C^-1')
        {
            const bool v = active && (s0 + lo) < B;
            const float *sol = SOL + (size_t)(v ? s0 + lo : 0) * C::NSOL;
            auto value = [&](int ks, int kk) __attribute__((always_inline)) {
                float val = 0.f;
                if (ks == 0) {
                    if (v && kk < KP) val = sol[kk];
                } else {
                    const int q = 32 * (ks - 1) + kk;
                    if (v && q < GX::KK2) val = sol[C::SOL_CI + q];
                }
                return val;
            };
            float xs[GX::NKS][8];                         // (read once: small batches are latency-bound)
#pragma unroll
            for (int ks = 0; ks < GX::NKS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) xs[ks][j] = value(ks, 8 * g + j);
            float sc0 = 1.f, sc1 = 1.f;
            if constexpr (GX::F16) {
                float m0 = 0.f, m1 = 0.f;
#pragma unroll
                for (int ks = 0; ks < GX::NKS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (ks == 0) m0 = fmaxf(m0, fabsf(xs[ks][j])); else m1 = fmaxf(m1, fabsf(xs[ks][j]));
                    }
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
                float i0, i1;
                sc0 = f16_row_scale(m0, i0);
                sc1 = f16_row_scale(m1, i1);
#pragma unroll
                for (int r = 0; r < 4; ++r) { is0[r] = __shfl(i0, 4 * g + r); is1[r] = __shfl(i1, 4 * g + r); }
            }
#pragma unroll
            for (int ks = 0; ks < GX::NKS; ++ks) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = xs[ks][j] * (ks == 0 ? sc0 : sc1);
                if constexpr (GX::F16) split8h(x, S1h[ks], S1m[ks]);
                else split8(x, S1h[ks], S1m[ks], S1l[ks]);
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
        auto offB_of = [&](int r) { return row_of(r) * (unsigned)Nb; };
        const float *abase = bt.A_blue ? bt.A_blue + (size_t)(active ? s0 : 0) * Nb : nullptr;      // (batch order: not indexed)
        // Spectrum s0 + r of the batch is row batch_row(s0 + r) of the batch arrays (ABI v3: rows / row_stride, qfa_common.h):
        // the wave's 16 row numbers wait in LDS (this role has no registers to spare) and every request forms its per-lane
        // 64-bit address from one of them
        unsigned *rtab = reinterpret_cast<unsigned *>(lds + GX::L_ROWS + w * 64);
        if (lane < 16) rtab[lane] = (unsigned)batch_row(bt, (active ? s0 : 0) + min(lane, last_row));
        const unsigned RS = (unsigned)bt.row_stride;                     // (elements; < 2^31: check_batch)
        // factored-z form: the per-spectrum factors of the wave's 16 spectra in LDS (this role has no registers to spare)
        float4 *zsl = reinterpret_cast<float4 *>(lds + GX::L_ZS + w * 256);
        if (ZF && lane < 16) zsl[lane] = (active && s0 + lane < B) ? ZS[s0 + lane] : float4{0.f, 0.f, 0.f, 0.f};
        // scalar-gradient sums: float32 inside a tile, float64 across tiles -- the float64 running sums live in LDS
        // (three doubles per lane), not in six registers
        double *scal = reinterpret_cast<double *>(lds + GX::L_SCAL) + (size_t)w * 3 * 64 + lane;
        scal[0] = 0.0; scal[64] = 0.0; scal[128] = 0.0;

        // ---- spectra of one tile for this wave: LDS-DMA of the 128-byte row segments of delta, sigma, zabs and of the
        // 32-byte row segments of the mask into the wave's staging buffer `par`.  Staging slot q of an array holds row
        // q ^ ((q >> 2) & 1): the four rows 4 g + r that one ds_read_b64 of take_tile touches then alternate between the
        // two halves of the banks (2-way = the minimum for 512 bytes).  16-byte pieces: lane = (slot = 8 i + (lane >> 3),
        // piece = lane & 7), two instructions per float array; the mask goes as 4-byte pieces, two instructions.
        // Returns the number of requests issued (wave-uniform; the counted wait of the NEXT step needs it), 0 = "wait
        // for everything":
        //    8   all 32 pixels inside the row (and inside the blue side, for a blue tile);
        //   14   the tile that straddles the end of the blue side: zabs as 4-byte pieces with clamped pixels;
        //    0   the ragged last tile of the pixel axis: everything as clamped 4-byte pieces, masks through registers.
        // The third array is staged for EVERY tile so that the counts are fixed: red tiles re-request their delta rows
        // there (hits in the vector cache, values unused).  No request reads past the end of a row.
        unsigned char *stg = lds + GX::L_STG + w * 2 * GX::STG_B;
        auto stage_tile = [&](int tg, int par) -> int {
            if (QFA_GX_ABL & 1) return 8;
            if (QFA_GX_ABL & 8) tg = t0;
            const bool zblue = !ZF && tg < nbt;                                               // wave-uniform
            const bool fastp = 32 * tg + 31 < Npix, fastz = ZF || !zblue || 32 * tg + 31 < Nb;
            // the third array: zabs rows (Nb apart, Nb long) on a blue tile, else the delta rows again
            const bool zreal = zblue && !(QFA_GX_ABL & 16);
            const float *zb = zreal ? bt.zabs : bt.delta;
            const unsigned zpitch = zreal ? (unsigned)Nb : RS;
            const int zlen = zreal ? Nb : Npix;
            const unsigned dst = wave_uniform(lds_addr(stg + par * GX::STG_B));
            auto slot_row = [&](int q) __attribute__((always_inline)) { return (unsigned)min(q ^ ((q >> 2) & 1), last_row); };
#if QFA_TRACKED_LOADS
            {   // test build: ordinary loads and LDS stores for all four arrays (two rows per pass), no counted wait
                float *sf = reinterpret_cast<float *>(stg + par * GX::STG_B);
                unsigned char *mb = stg + par * GX::STG_B + GX::STG_MASK;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = 2 * i + (lane >> 5), pxl = lane & 31;
                    const unsigned R = rtab[slot_row(q)];
                    const int px = 32 * tg + pxl;
                    const unsigned long long o = (unsigned long long)R * RS + (unsigned)min(px, Npix - 1);
                    sf[0 * (GX::STG_ARR / 4) + q * 32 + pxl] = bt.delta[o];
                    sf[1 * (GX::STG_ARR / 4) + q * 32 + pxl] = bt.error[o];
                    if (!ZF) sf[2 * (GX::STG_ARR / 4) + q * 32 + pxl] = zb[(unsigned long long)R * zpitch + (unsigned)min(px, zlen - 1)];
                    mb[q * 32 + pxl] = px < Npix ? bt.mask[o] : (unsigned char)0;
                }
                (void)dst; (void)fastz;
                return 0;
            }
#endif
            if (fastp) {
                unsigned R2[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) R2[i] = rtab[slot_row(8 * i + (lane >> 3))];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned pc = 32u * (unsigned)tg + 4u * (unsigned)(lane & 7);             // first pixel of the piece
                    asm volatile("" : "+v"(pc));
                    const unsigned long long o = (unsigned long long)R2[i] * RS + pc;
                    glds16p(lane_ptr<2>(bt.delta, o), dst + 0 * GX::STG_ARR + i * 1024);
                    glds16p(lane_ptr<2>(bt.error, o), dst + 1 * GX::STG_ARR + i * 1024);
                    if (!ZF && fastz) glds16p(lane_ptr<2>(zb, (unsigned long long)R2[i] * zpitch + pc), dst + 2 * GX::STG_ARR + i * 1024);
                    glds4p(lane_ptr<0>(bt.mask, o), dst + GX::STG_MASK + i * 256);
                }
                if (fastz) return ZF ? 6 : 8;
            }
            // 4-byte pieces, 64 lanes = two rows per instruction, the pixel index clamped per lane (at most two tiles of a
            // work item come here: rolled)
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                const int q = 2 * i + (lane >> 5);
                const unsigned R = rtab[slot_row(q)];
                int pxl = lane & 31;
                asm volatile("" : "+v"(pxl));
                if (!ZF) glds4p(lane_ptr<2>(zb, (unsigned long long)R * zpitch + (unsigned)min(32 * tg + pxl, zlen - 1)), dst + 2 * GX::STG_ARR + i * 256);
                if (!fastp) {
                    const unsigned long long o = (unsigned long long)R * RS + (unsigned)min(32 * tg + pxl, Npix - 1);
                    glds4p(lane_ptr<2>(bt.delta, o), dst + 0 * GX::STG_ARR + i * 256);
                    glds4p(lane_ptr<2>(bt.error, o), dst + 1 * GX::STG_ARR + i * 256);
                }
            }
            if (fastp) return 14;
            // masks of the ragged tile: ordinary loads (hipcc waits for them by itself) and byte stores
            unsigned char *mb = stg + par * GX::STG_B + GX::STG_MASK;
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                const int q = 2 * i + (lane >> 5);
                const unsigned R = rtab[slot_row(q)];
                const int px = 32 * tg + (lane & 31);
                mb[q * 32 + (lane & 31)] = px < Npix ? bt.mask[(unsigned long long)R * RS + (unsigned)px] : (unsigned char)0;
            }
            return 0;
        };
        // copy the lane's 4 spectra x 2 pixels of the staged tile into registers; the mask goes into the sign of sigma
        // (only sigma^2 is ever used: sign bit set means "masked" from here on)
        auto take_tile = [&](int par, SpecA &cur) {
            const unsigned char *sb = stg + par * GX::STG_B + 8 * lo;
            const unsigned char *mb = stg + par * GX::STG_B + GX::STG_MASK + 2 * lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int slot = 4 * g + (r ^ (g & 1));
                cur.d[r] = *reinterpret_cast<const f32x2 *>(sb + 0 * GX::STG_ARR + slot * 128);
                f32x2 e2 = *reinterpret_cast<const f32x2 *>(sb + 1 * GX::STG_ARR + slot * 128);
                if (!ZF) cur.z[r] = *reinterpret_cast<const f32x2 *>(sb + 2 * GX::STG_ARR + slot * 128);
                const unsigned mk = *reinterpret_cast<const unsigned short *>(mb + slot * 32);
                e2[0] = (mk & 0xffu) ? fabsf(e2[0]) : -1.f;          // (sign bit, not value: sigma = +0 stays unmasked)
                e2[1] = (mk & 0xff00u) ? fabsf(e2[1]) : -1.f;
                cur.sg[r] = e2;
            }
        };

        // one 16-pixel half of a tile: stage 1 (36 MFMAs on the ring slot of this half; the per-pixel parameters of the half
        // are read with it), then stage 2 of the lane's four elements.  (A form that issued stage 1 of half t + 1 beside stage 2
        // of half t inside the wave -- sched_group_barrier interleave, image requested two half-steps ahead -- was built and
        // measured in round 3: same results, 2.67 against 2.62 ms at c3: the SIMD issues about one instruction per 4.8 cycles
        // over BOTH waves, so ordering the streams inside a wave buys nothing; profiles/r3_ablation_pass2.txt.)
        struct PixA {
            float Psi, om, ti, pwi, l2i;
        };
        float t_tau0 = 0.f, t_c0 = 0.f, t_beta = 0.f;
        auto stage1A = [&](int slot, f32x4 &ofy, f32x4 &oq, PixA &pp) {
            const unsigned char *img = lds + GX::L_IMG + slot * GX::HALF_B;
            const unsigned char *bp = img + lane * 16;
            f32x4 afy = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f};
            constexpr int NP = GX::NP;
            u32x4 bq[2][NP];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) bq[0][pc] = *reinterpret_cast<const u32x4 *>(bp + pc * 1024);
#pragma unroll
            for (int ks = 0; ks < GX::NKS; ++ks) {
                if (ks + 1 < GX::NKS) {
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc)
                        bq[(ks + 1) & 1][pc] = *reinterpret_cast<const u32x4 *>(bp + (ks + 1) * GX::KS_B + pc * 1024);
                }
                const u32x4 &bh = bq[ks & 1][0], &bm = bq[ks & 1][1], &bl = bq[ks & 1][NP - 1];
                if constexpr (GX::F16) {
                    if (ks == 0) afy = xdl3h(S1h[ks], S1m[ks], bh, bm, afy);
                    else aq = xdl3h(S1h[ks], S1m[ks], bh, bm, aq);
                } else {
                    constexpr int L0 = GX::F16 ? 0 : 1;
                    if (ks == 0) afy = xdl6(S1h[ks], S1m[ks], S1l[L0 * ks], bh, bm, bl, afy);
                    else aq = xdl6(S1h[ks], S1m[ks], S1l[L0 * ks], bh, bm, bl, aq);
                }
            }
            const float *po = reinterpret_cast<const float *>(img + GX::S1_HALF);
            if constexpr (GX::F16) {           // the powers of two back in: element r <-> spectrum 4 g + r, the lane's pixel
                const float it1 = po[GX::PAR_IT1 + lo], it2 = po[GX::PAR_IT2 + lo];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    afy[r] = (afy[r] * is0[r]) * it1;
                    aq[r] = (aq[r] * is1[r]) * it2;
                }
            }
            pp.Psi = po[lo]; pp.om = po[16 + lo];
            pp.ti = pp.pwi = pp.l2i = 0.f;
            if (ZF) { pp.ti = po[32 + lo]; pp.pwi = po[48 + lo]; pp.l2i = po[64 + lo]; }
            ofy = afy;
            oq = aq;
        };
        auto stage2A = [&](auto blue_tag, int tg, int h, const SpecA &cur, int par, const f32x4 &afy, const f32x4 &aq,
                           const PixA &pp) {
            constexpr bool BLUE = decltype(blue_tag)::value;
            float *bslot = reinterpret_cast<float *>(lds + GX::L_BETA + (par * GX::NG + w) * 2048);
            float *gslot = reinterpret_cast<float *>(lds + GX::L_GAM + (par * GX::NG + w) * 32 * GX::GROW * 4);
            float *psum = reinterpret_cast<float *>(lds + GX::L_PSUM + (par * GX::NG + w) * 512);
            const float Psi = pp.Psi, om = pp.om;
            const float ti = pp.ti, pwi = pp.pwi, l2i = pp.l2i;
            const int px = 32 * tg + 2 * lo + h;
            const bool inb = px < Npix;
            const bool blue = px < Nb;
            float gamR[4];
            float gPsi = 0.f, gOm = 0.f, sA = 0.f, cnt = 0.f;
            float betaR[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool wv_ = inb & sv[r] & (__float_as_int(cur.sg[r][h]) >= 0);
                float dd = wv_ ? cur.d[r][h] : 0.f;
                const float sg = cur.sg[r][h];
                if (BLUE) {
                    float l2, pw, Ab, re;
                    if (ZF) {                                                                         // qfa_common.h, ZFac
                        const float4 zq = zsl[4 * g + r];
                        l2 = zq.z + l2i;
                        pw = zq.y * pwi;
                        Ab = fast_exp2(fmaf(zq.x, ti, k.offp));                                       // QFA/model.py:125
                        re = k.omc0 - fast_exp2(k.k1 * pw);                                           // QFA/utils.py:91
                    } else {
                        l2 = fast_log2(1.0f + cur.z[r][h]);
                        pw = fast_exp2(k.beta * l2);
                        const float tauv = k.t_amp * fast_exp2(k.t_expo * (l2 + k.t_lscale)) + k.t_off;   // QFA/utils.py:105-141
                        Ab = fast_exp2(-tauv * QFA_LOG2E);                                            // QFA/model.py:125
                        re = 1.0f - k.c0 - fast_exp2(-k.tau0 * pw * QFA_LOG2E);                       // QFA/utils.py:91
                    }
                    if (HASA) Ab = abase[offB_of(r) + (unsigned)min(px, Nb - 1)];                     // custom tau callable
                    const float Av = blue ? Ab : 1.f;
                    const float zd = blue ? re * re : 0.f;
                    const float A2 = Av * Av;
                    const float D = A2 * Psi + om * zd + sg * sg;
                    const float wD = wv_ ? fast_rcp(D) : 0.f;
                    const float wDA = wD * Av;
                    const float uu = wD * (dd - Av * afy[r]);                   // (Sigma^-1 delta)_i
                    const float dS = wD - wDA * wDA * aq[r];                    // diag(Sigma^-1)_i
                    const float dG = 0.5f * (dS - uu * uu);                     // QFA/model.py:136,138
                    gPsi += A2 * dG;                                            // :139
                    gOm += dG * zd;                                             // :140
                    const float root = 1.0f - k.tau0 * pw - k.c0;               // :141
                    const float e = dG * (om * zd) * zd * 2.0f * root;
                    t_tau0 -= e * pw;                                           // :142
                    t_beta -= e * (k.tau0 * pw * (l2 * QFA_LN2));               // :143
                    t_c0 -= e;                                                  // :144
                    cnt += wv_ ? 1.f : 0.f;
                    betaR[r] = wDA * Av;
                    sA += betaR[r] * Av;
                    gamR[r] = Av * uu;
                } else {                                                        // red side: A = 1, zd = 0
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
                if (r & 1) __builtin_amdgcn_sched_barrier(0);      // two elements at a time: bounds the live temporaries
            }
            GXS((BLUE ? 0 : 16) + 8 * h + 2)
            if constexpr (TERMS == 6) {
                // W form of stage 3 (role B below): beta and gamma of the lane's four spectra stay in THIS lane's layout --
                // [half h][lane][4 floats], one 16-byte store each
                *reinterpret_cast<float4 *>(bslot + (h * 64 + lane) * 4) = float4{betaR[0], betaR[1], betaR[2], betaR[3]};
                *reinterpret_cast<float4 *>(gslot + (h * 64 + lane) * 4) = float4{gamR[0], gamR[1], gamR[2], gamR[3]};
            } else {
            // beta[s = 4g + r][pxl = 2 lo + h]
#pragma unroll
            for (int r = 0; r < 4; ++r) bslot[(4 * g + r) * 32 + 2 * lo + h] = betaR[r];
            // gamma transposed: row rho = 16 h + lo, columns s = 4g .. 4g + 3 (one 16-byte store)
            *reinterpret_cast<float4 *>(gslot + (16 * h + lo) * GX::GROW + 4 * g) =
                float4{gamR[0], gamR[1], gamR[2], gamR[3]};
            }
            // per-pixel sums over the wave's 16 spectra (lanes lo + 16 g'): two cross-lane adds, one store per pixel
            // (v_permlane16_swap / v_permlane32_swap: three exchanges and three adds leave quantity g's sum over the
            // four 16-lane rows in row g)
            {
                const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sA), __float_as_uint(gPsi), false, false);
                const auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(gOm), __float_as_uint(cnt), false, false);
                const float u01 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);   // rows: sA(0+1) gPsi(0+1) sA(2+3) gPsi(2+3)
                const float u23 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);   //       gOm     cnt       gOm     cnt
                const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(u01), __float_as_uint(u23), false, false);
                // t[0] rows: sA(0+1) gPsi(0+1) gOm(0+1) cnt(0+1);  t[1] rows: sA(2+3) gPsi(2+3) gOm(2+3) cnt(2+3)
                psum[g * 32 + 2 * lo + h] = __uint_as_float(t[0]) + __uint_as_float(t[1]);
            }
            if (BLUE && h == 1) {
                scal[0] += (double)t_tau0;
                scal[64] += (double)t_c0;
                scal[128] += (double)t_beta;
                t_tau0 = 0.f; t_c0 = 0.f; t_beta = 0.f;
            }
            GXS((BLUE ? 0 : 16) + 8 * h + 3)
        };

        // Tile c is staged in buffer c & 1, which is refilled for tile c + 2 during step c (after the copy-out at the
        // start of the step): the requests have more than a step and a half to land.
        int cnt_a = 0, cnt_b = 0;            // requests in flight for buffer 0 / 1 (0: not issued by a counted path)
        if (n > 0 && active) {
            cnt_a = stage_tile(tile_of(0), 0);
            if (n > 1) cnt_b = stage_tile(tile_of(1), 1);
        }
        step_barrier();                      // (role B's wait in front of this barrier covers the first image half)
        SpecA cur;
        f32x4 afy = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f};
        PixA pxp{0.f, 0.f, 0.f, 0.f, 0.f};
        auto tileA = [&](int c, int &cnt_cur, int cnt_other) {
            const bool work = c < n && active;
            const int tg = work ? tile_of(c) : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (work) {
                    auto spectra = [&]() {
                        if (h != 0) return;
                        // this tile's staged spectra: everything but the requests of tile c + 1 has landed after the wait;
                        // copy out, then the requests for tile c + 2 into the buffer just read
                        if (QFA_GX_ABL & 1) {}
                        else if (c + 1 < n && cnt_other == 8) dma_wait<8>();
                        else if (c + 1 < n && cnt_other == 6) dma_wait<6>();
                        else if (c + 1 < n && cnt_other == 14) dma_wait<14>();
                        else dma_wait<0>();
                        take_tile(c & 1, cur);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // staging buffer read: it may be overwritten now
                        __builtin_amdgcn_sched_barrier(0);
                        cnt_cur = 0;
                        if (c + 2 < n) cnt_cur = stage_tile(tile_of(c + 2), c & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    {
                        stage1A(h, afy, aq, pxp);
                        __builtin_amdgcn_sched_barrier(0);
                        spectra();                             // (behind stage 1 of the first half, which needs no spectra)
                        if (tg < nbt) stage2A(std::true_type{}, tg, h, cur, c & 1, afy, aq, pxp);
                        else stage2A(std::false_type{}, tg, h, cur, c & 1, afy, aq, pxp);
                    }
                }
                step_barrier();
            }
        };
        for (int c = 0; c < n + 2; c += 2) {
            tileA(c, cnt_a, cnt_b);
            if (c + 1 < n + 2) tileA(c + 1, cnt_b, cnt_a);
        }
#if QFA_GX_STAMPS
        if (blockIdx.x == 300 && w == 0 && lane == 0) {
            st_[30] = n;
            for (int i = 0; i < 32; ++i) qfa_gx_stamps[i] = st_[i];
        }
#endif
        dma_wait<0>();
        double s_tau0 = scal[0], s_c0 = scal[64], s_beta = scal[128];
        if (active) {
            for (int o = 32; o >= 1; o >>= 1) {
                s_tau0 += __shfl_xor(s_tau0, o);
                s_c0 += __shfl_xor(s_c0, o);
                s_beta += __shfl_xor(s_beta, o);
            }
            if (lane == 0) {
                if (det) {
                    double *q = slabS + ((size_t)blockIdx.x * GX::NG + w) * 3;
                    q[0] = s_tau0; q[1] = s_c0; q[2] = s_beta;
                } else {
                    scal64_commit(sc64, s_tau0, s_c0, s_beta, gridDim.x * (unsigned)GX::NG, accS);
                }
            }
        } else if (det && lane == 0) {
            double *q = slabS + ((size_t)blockIdx.x * GX::NG + w) * 3;
            q[0] = 0.0; q[1] = 0.0; q[2] = 0.0;
        } else if (!det) {
            scal64_commit(sc64, 0.0, 0.0, 0.0, gridDim.x * (unsigned)GX::NG, accS);
        }
    } else if (QFA_GX_ROLE != 1) {
        // ================================================================ role B: image DMA, flushes, stage 3
        constexpr bool WB = TERMS == 6;        // stage 3 in its W form (float32 grade); TERMS == 3: the G form (QFA_F_S3_FAST)
        // ---- W form: accF[px][b] = sum_a F[px][a] W[px][a][b], W = sum_s Z_s[a][b] beta[s][px]
        // as a K = spectrum GEMM per column tile a: A = static Z pieces (row m = b = lane & 15; k = 8 g + j <-> spectrum
        // 4 g + (j & 3), piece slot j >> 2), B = the lane's own four beta values as {h|l}, {m|m}, {h|h}: three INDEPENDENT
        // short chains per column tile instead of six dependent 32x32x16 MFMAs per spectrum pair -- role B alone ran 2.75 ms
        // at c3 in the six-product G form (latency of the dependent chain), against 1.1 ms with three.
        const int loB = lane & 15, gB = lane >> 4;
        u32x4 ZA1[WB ? KP : 1], ZA2[WB ? KP : 1], PA1 = {0u, 0u, 0u, 0u}, PA2 = {0u, 0u, 0u, 0u};
        if constexpr (WB) {
            const float *solr[4];
            bool vr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sidx = s0 + 4 * gB + r;
                vr[r] = active && sidx < B && loB < Nh && loB < KP;
                solr[r] = SOL + (size_t)(vr[r] ? sidx : 0) * C::NSOL;
            }
#pragma unroll
            for (int a = 0; a < KP; ++a) {
                float x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = vr[r] ? solr[r][C::SOL_Z + a * KP + (loB & (KP - 1))] : 0.f;
                unsigned h01, m01, l01, h23, m23, l23;
                split2(x[0], x[1], h01, m01, l01);
                split2(x[2], x[3], h23, m23, l23);
                ZA1[a] = u32x4{l01, l23, h01, h23};
                ZA2[a] = u32x4{h01, h23, m01, m23};
            }
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = vr[r] ? solr[r][C::SOL_P + (loB & (KP - 1))] : 0.f;
            unsigned h01, m01, l01, h23, m23, l23;
            split2(x[0], x[1], h01, m01, l01);
            split2(x[2], x[3], h23, m23, l23);
            PA1 = u32x4{l01, l23, h01, h23};
            PA2 = u32x4{h01, h23, m01, m23};
        }
        // ---- G form (TERMS == 3)
        // One stage-3 MFMA (32 px x 32 columns x K = 16) covers the columns of 32 / KP spectra: a pair at KP = 16, four
        // spectra at KP = 8 (where only k < 8 carries data).  Column col = (spectrum sc = col / KP, b = col % KP).
        constexpr int SPM = 32 / KP, NMG = 16 / SPM;       // spectra per MFMA, MFMA groups per wave (8 pairs / 4 fours)
        const int col = lane & 31, h2 = lane >> 5, b = lane & (KP - 1), sc = col / KP, sp = (lane >> 4) & 1;
        const int tidB = tid & 255;                        // 0..255 over the four role-B waves
        // B operands: Z of group m: B[k = a = 8 h2 + j][col = (sc, b)] = Z_{SPM m + sc}[a][b]
        constexpr int NMGA = WB ? 1 : NMG;
        u32x4 Zh[NMGA], Zm[NMGA], Zl[NMGA], Ph = {0u, 0u, 0u, 0u}, Pm = Ph, Pl = Ph;
        if constexpr (!WB) {
#pragma unroll
            for (int m = 0; m < NMG; ++m) {
                const int s = s0 + SPM * m + sc;
                const bool v = active && s < B && b < Nh;
                const float *sol = SOL + (size_t)(v ? s : 0) * C::NSOL + C::SOL_Z + b;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = (v && 8 * h2 + j < KP) ? sol[(8 * h2 + j) * KP] : 0.f;
                split8(x, Zh[m], Zm[m], Zl[m]);
            }
            // gamma term: B[k = s = 8 h2 + j][col] = p_s[b] for col < KP, 0 otherwise
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = s0 + 8 * h2 + j;
                const bool v = active && s < B && col < KP && b < Nh;
                x[j] = v ? SOL[(size_t)s * C::NSOL + C::SOL_P + b] : 0.f;
            }
            split8(x, Ph, Pm, Pl);
        }
        // LDS-DMA of half-step t = 2 c + h: the stage-1 image + Psi/omega of that half into ring slot h, and with the
        // second half the F pieces of the tile into the F ring; wave w moves the 1-KiB pieces w, w + 4, ...
        // (role A runs stage 1 of half t + 1 during half-step t: the image of half t + 2 is requested at the start of
        // half-step t into the slot half t was read from in half-step t - 1; the F block of tile c with its second half-step)
        auto get_img = [&](int t) {
            if ((QFA_GX_ABL & 4) && t > 1) return;
            const int c = t >> 1, h = t & 1;
            const unsigned char *sbase = uniform_ptr(PGX + (size_t)tile_of(c) * GX::TILE_B);
            unsigned char *img = lds + GX::L_IMG + h * GX::HALF_B;
#pragma unroll
            for (int i = 0; i < (GX::NCH_HALF + GX::NG - 1) / GX::NG; ++i) {
                const int ch = w + GX::NG * i;
                if (ch < GX::NCH_HALF)
                    glds16a(sbase + h * GX::HALF_B + ch * 1024, (unsigned)lane * 16u, wave_uniform(lds_addr(img + ch * 1024)));
            }
        };
        auto get_F = [&](int c) {
            if ((QFA_GX_ABL & 4) && c > 0) return;
            const unsigned char *sbase = uniform_ptr(PGX + (size_t)tile_of(c) * GX::TILE_B);
            unsigned char *fp = lds + GX::L_FP + (c & 1) * 3072;
            if (w < 3) glds16a(sbase + GX::OFF_FP + w * 1024, (unsigned)lane * 16u, wave_uniform(lds_addr(fp + w * 1024)));
        };
        // Every lane of a flushing wave issues its request: the number of requests per wave is then a constant, which
        // the counted wait below needs.  Default mode (float atomics): a lane outside the arrays adds 0 to an element
        // inside them, a different one for each lane (195 000 tile-steps adding to ONE spare address took 50 ms).
        // Deterministic mode (plain stores into the block's slab row, every element written exactly once): such a lane
        // stores into the 64 spare floats at the end of the row.
        float *sink = accF + (slab_stride - 64) + lane;
        // tile tg leaves the workgroup: role B sums the four groups' partials (fixed order) and adds them to the packed
        // buffer.  Default: 256 threads, 32 KP / 256 outputs each (a wave's 64 lanes cover 256 contiguous bytes at
        // N_h = KP).  Deterministic with N_h a multiple of 4: the first 8 KP threads, one 16-byte store each.
        const bool wide = det && (Nh & 3) == 0;
        constexpr int NWIDE = 8 * KP;                      // threads of the 16-byte form: 128 (waves 0, 1) / 64 (wave 0)
        auto flush_F = [&](int tg, int par) {
            if (QFA_GX_ABL & 2) return;
            const float *pp = reinterpret_cast<const float *>(lds + GX::L_PART + par * GX::NG * GX::PARTF * 4);
            if (wide) {
                if (tidB >= NWIDE) return;                                            // wave-uniform
                const int pxl = tidB / (KP / 4), b4 = 4 * (tidB % (KP / 4));
                const int px = 32 * tg + pxl;
                const float4 *q4 = reinterpret_cast<const float4 *>(pp + pxl * KP + b4);
                const float4 v0 = q4[0], v1 = q4[GX::PARTF / 4], v2 = q4[2 * GX::PARTF / 4], v3 = q4[3 * GX::PARTF / 4];
                const float4 v = {(v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y),
                                  (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w)};
                const bool ok = (b4 < Nh) & (px < Npix);
                if (ok) *reinterpret_cast<float4 *>(accF + (size_t)px * Nh + b4) = v;
                else *sink = v.x;
                return;
            }
#pragma unroll
            for (int k4 = 0; k4 < GX::PARTF / 256; ++k4) {
                const int o = tidB + 256 * k4;
                float v = (QFA_GX_ABL & 64) ? 1.f : (pp[o] + pp[GX::PARTF + o]) + (pp[2 * GX::PARTF + o] + pp[3 * GX::PARTF + o]);
                const int px = 32 * tg + o / KP, bb = o % KP;
                const bool ok = (bb < Nh) & (px < Npix);
                if (QFA_GX_ABL & 32) { asm volatile("" ::"v"(v)); continue; }
                if (det) *(ok ? accF + (size_t)px * Nh + bb : sink) = v;
                else atomicAdd(accF + (size_t)min(px, Npix - 1) * Nh + bb % Nh, ok ? v : 0.f);
            }
        };
        // per-pixel sums [sumA | gPsi | gOmega | cnt] of tile tg: thread (which = t >> 5, pxl = t & 31) of 128 -- waves
        // 0 and 1, or waves 2 and 3 when the F sums go out as 16-byte stores (one request per wave and tile then)
        auto flush_P = [&](int tg, int par) {
            if (QFA_GX_ABL & 2) return;
            if (wide ? tidB < 128 : tidB >= 128) return;                              // wave-uniform (waves 2, 3 / 0, 1)
            const int which = (tidB >> 5) & 3, pxl = tidB & 31;
            const float *q = reinterpret_cast<const float *>(lds + GX::L_PSUM + par * GX::NG * 512) + which * 32 + pxl;
            float v = (QFA_GX_ABL & 64) ? 1.f : (q[0] + q[128]) + (q[256] + q[384]);
            if (QFA_GX_ABL & 32) { asm volatile("" ::"v"(v)); return; }
            const int px = 32 * tg + pxl;
            const bool ok = (px < Npix) & ((which != 2) | (px < Nb));
            // (default mode: a red pixel's lane of the gOmega group adds 0 to the pixel's count instead)
            const int pxc = min(px, Npix - 1);
            const int offc = (which == 2 && pxc >= Nb) ? 2 * Npix + Nb + pxc : which * Npix - (which == 3 ? Npix - Nb : 0) + pxc;
            if (det) *(ok ? accA + offc : sink) = v;
            else atomicAdd(accA + offc, ok ? v : 0.f);
        };
        // stage 3 of tile c, in two parts (the two half-steps of tile c + 1; balanced, so that neither half-step waits for
        // this role): part 0 = the gamma term and the first half of the MFMA groups, part 1 = the second half, added to
        // part 0's sums in LDS (the accumulator does not live across the barrier: this role is at the register limit)
        auto tileB = [&](int c, auto part_tag) {
            constexpr int PART = decltype(part_tag)::value;
            const int par = c & 1;
            if constexpr (WB) {
                // half PART of tile c: beta / gamma of the lane's four spectra (pixel 2 lo + PART) as role A left them
                const float *bsl = reinterpret_cast<const float *>(lds + GX::L_BETA + (par * GX::NG + w) * 2048);
                const float *gsl = reinterpret_cast<const float *>(lds + GX::L_GAM + (par * GX::NG + w) * 32 * GX::GROW * 4);
                const float4 b4 = *reinterpret_cast<const float4 *>(bsl + (PART * 64 + lane) * 4);
                const float4 g4 = *reinterpret_cast<const float4 *>(gsl + (PART * 64 + lane) * 4);
                const float *frow = reinterpret_cast<const float *>(lds + GX::L_FP + (c & 1) * 3072) + (2 * loB + PART) * GX::FROW;
                float fa[KP];
#pragma unroll
                for (int a4 = 0; a4 < KP / 4; ++a4) {
                    const float4 f4 = *reinterpret_cast<const float4 *>(frow + 4 * a4);
                    fa[4 * a4] = f4.x; fa[4 * a4 + 1] = f4.y; fa[4 * a4 + 2] = f4.z; fa[4 * a4 + 3] = f4.w;
                }
                unsigned h01, m01, l01, h23, m23, l23;
                split2(b4.x, b4.y, h01, m01, l01);
                split2(b4.z, b4.w, h23, m23, l23);
                const u32x4 bhl = {h01, h23, l01, l23}, bmm = {m01, m23, m01, m23}, bhh = {h01, h23, h01, h23};
                split2(g4.x, g4.y, h01, m01, l01);
                split2(g4.z, g4.w, h23, m23, l23);
                const u32x4 ghl = {h01, h23, l01, l23}, gmm = {m01, m23, m01, m23}, ghh = {h01, h23, h01, h23};
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                f32x4 acc = xdl(PA2, ghh, xdl(PA2, gmm, xdl(PA1, ghl, zero)));       // sum_s p_s[b] gamma[s][px]
                // column tiles in chunks of four: twelve MFMAs in flight (independent chains of three), the FMAs of a chunk
                // behind the MFMAs of the next
                f32x4 Wc[2][4];
#pragma unroll
                for (int ch = 0; ch <= KP / 4; ++ch) {
                    if (ch < KP / 4) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int a = 4 * ch + j;
                            Wc[ch & 1][j] = xdl(ZA2[a], bhh, xdl(ZA2[a], bmm, xdl(ZA1[a], bhl, zero)));
                        }
                    }
                    if (ch > 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int a = 4 * (ch - 1) + j;
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[r] = fmaf(fa[a], Wc[(ch - 1) & 1][j][r], acc[r]);
                        }
                    }
                }
                // the lane holds accF[px = 2 lo + PART][b = 4 g + r] of its group: one 16-byte store into the group's slot
                float *part = reinterpret_cast<float *>(lds + GX::L_PART + (par * GX::NG + w) * GX::PARTF * 4);
                if (KP == 16 || gB < KP / 4)
                    *reinterpret_cast<float4 *>(part + (2 * loB + PART) * KP + 4 * gB) = float4{acc[0], acc[1], acc[2], acc[3]};
                return;
            }
            const unsigned char *fp = lds + GX::L_FP + (c & 1) * 3072 + lane * 16;
            const float *bslot = reinterpret_cast<const float *>(lds + GX::L_BETA + (par * GX::NG + w) * 2048);
            const float *gslot = reinterpret_cast<const float *>(lds + GX::L_GAM + (par * GX::NG + w) * 32 * GX::GROW * 4);
            float *part = reinterpret_cast<float *>(lds + GX::L_PART + (par * GX::NG + w) * GX::PARTF * 4);
            const u32x4 Fh = *reinterpret_cast<const u32x4 *>(fp), Fm = *reinterpret_cast<const u32x4 *>(fp + 1024),
                        Fl = *reinterpret_cast<const u32x4 *>(fp + 2048);
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;
            f32x16 acc = zero;
            if (PART == 0) {
                // gamma of pixel pxl = col (row rho = 16 (col & 1) + (col >> 1)), spectra 8 h2 .. 8 h2 + 7
                const float *grow = gslot + (16 * (col & 1) + (col >> 1)) * GX::GROW + 8 * h2;
                const float4 g0 = *reinterpret_cast<const float4 *>(grow), g1 = *reinterpret_cast<const float4 *>(grow + 4);
                const float gx[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                u32x4 Gh, Gm, Gl;
                split8(gx, Gh, Gm, Gl);
                acc = xdl32_6<TERMS>(Gh, Gm, Gl, Ph, Pm, Pl, zero);
            }
#pragma unroll
            for (int m = (NMG / 2) * PART; m < (NMG / 2) * (PART + 1); ++m) {
                const f32x16 G = xdl32_6<TERMS>(Fh, Fm, Fl, Zh[m], Zm[m], Zl[m], zero);
                const float *brow = bslot + (SPM * m + sc) * 32 + 4 * h2;       // pixels 8 q + 4 h2 + (0..3)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bq = *reinterpret_cast<const float4 *>(brow + 8 * q);
                    acc[4 * q + 0] = fmaf(bq.x, G[4 * q + 0], acc[4 * q + 0]);
                    acc[4 * q + 1] = fmaf(bq.y, G[4 * q + 1], acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(bq.z, G[4 * q + 2], acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(bq.w, G[4 * q + 3], acc[4 * q + 3]);
                }
            }
            // Sum the spectra of the groups (the lanes that share b).  Lanes l and l ^ 16: v_permlane16_swap exchanges
            // the odd 16-lane rows of acc[i] with the even rows of acc[8 + i], so one add leaves the sums of acc[i] in
            // the even rows (sp = 0) and those of acc[8 + i] in the odd rows.  KP = 8: also lanes l and l ^ 8 (a rotation
            // by 8 inside the row, DPP).  Each half of the lanes then stores half the pixel rows:
            // part[pxl][b], pxl = (ii & 3) + 8 (ii >> 2) + 4 h2 with ii = i + 8 sp
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i]), __float_as_uint(acc[8 + i]), false, false);
                float v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                if (KP == 8)
                    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
                const int pxl0 = (i & 3) + 8 * (i >> 2) + 4 * h2;                 // ii = i; ii = 8 + i adds 16 pixels
                float *q = part + (pxl0 + 16 * sp) * KP + b;
                if (KP == 16 || (lane & 8) == 0) {
                    if (PART == 0) *q = v;
                    else *q += v;             // (read-add-write: ds_add_f32 cost 2 500 cycles more per tile)
                }
            }
        };
#if QFA_GX_BPRIO && QFA_GX_BPRIO < 4
        __builtin_amdgcn_s_setprio(QFA_GX_BPRIO);
#endif
        if (n > 0) get_img(0);
        dma_wait<0>();
        step_barrier();
        for (int c = 0; c < n + 2; ++c) {
#if QFA_GX_BPRIO == 4
            // this role is the slower one while role A works on a red tile, and the faster one on a blue tile
            if (c < n && tile_of(c) < nbt) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = 2 * c + h;
                // The image DMA first, the flushes behind it, and a wait that leaves exactly the flushes in flight: they
                // are device-scope atomics with a long round trip, and have until the end of the NEXT half-step.
                if (t + 1 < 2 * n) get_img(t + 1);
                if (h == 1 && c < n) get_F(c);
                // requests of this wave's flushes (wave-uniform): F 2 (one as a 16-byte store, waves 0 and 1), P 1
                const bool wP = wide ? tidB >= 128 : tidB < 128, wF = !wide || tidB < NWIDE;
                const int nreq = (h == 0 && c >= 1 && c <= n && wP ? 1 : 0) +
                                 (h == 0 && c >= 2 && wF ? (wide ? 1 : GX::PARTF / 256) : 0);
                if (h == 0) {
                    if (c >= 1 && c <= n) flush_P(tile_of(c - 1), (c - 1) & 1);
                    if (c >= 2) flush_F(tile_of(c - 2), c & 1);
                }
                GXS(8 * h + 0)
                if (c >= 1 && c <= n && active) {
                    if (h == 0) tileB(c - 1, std::integral_constant<int, 0>{});
                    else tileB(c - 1, std::integral_constant<int, 1>{});
                }
                GXS(8 * h + 1)
                if (QFA_GX_ABL & 2) dma_wait<0>();
                else if (nreq == 3) dma_wait<3>();
                else if (nreq == 2) dma_wait<2>();
                else if (nreq == 1) dma_wait<1>();
                else dma_wait<0>();
                GXS(8 * h + 2)
                step_barrier();
                GXS(8 * h + 3)
            }
        }
#if QFA_GX_STAMPS
        if (blockIdx.x == 300 && w == 0 && lane == 0) {
            st_[30] = n;
            for (int i = 0; i < 32; ++i) qfa_gx_stamps[32 + i] = st_[i];
        }
#endif
    }
}
