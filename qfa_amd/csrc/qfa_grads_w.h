// qfa_grads_w.h -- pass 2 (gradients) for N_h <= 16, round-3 form: ONE wave per SIMD does stages 1, 2 and 3 of its 16
// spectra, every contraction on the bf16 XDL pipe at float32 grade (six piece products), nothing handed between waves.
//
// Why a third form (k_grads_x of qfa_grads_x.h keeps two ROLES per SIMD; k_grads the float32 MFMA):
//   k_grads_x's tile step is ~7 800 cycles against ~2 000 of XDL and ~2 800 of VALU time per SIMD: two dependent chains
//   (MFMA -> VALU -> LDS -> barrier) coupled by two barriers per tile, beta / gamma travelling role A -> LDS -> role B, the
//   per-tile partial sums of eight waves meeting in LDS, and no third wave to fill the stalls (231 VGPRs per wave).
// Here stage 3 is re-associated so that it needs NO transposition of the stage-2 results:
//     accF[px][b] = sum_s beta[s][px] (F_tile Z_s)[px][b]  =  sum_a F[px][a] W[px][a][b],   W[px][a][b] = sum_s Z_s[a][b] beta[s][px]
//   W^T is a GEMM with M = b (16 per column tile a), N = px (16), K = spectrum: its B operand is the lane's OWN four beta
//   values (lane (px, g) of the stage-1 output holds the spectra 4 g + r: exactly B[k = 8 g + j][n = px] of
//   v_mfma_f32_16x16x32_bf16 when k = 8 g + j stands for (spectrum 4 g + (j & 3), piece slot j >> 2)), its A operand
//   the static Z pieces.  The two piece slots of K = 32 carry two of the six piece products per MFMA:
//       A = {Zl | Zh} x B = {bh | bl},   A = {Zh | Zm} x B = {bm | bm},   A = {Zh | Zm} x B = {bh | bh}
//   i.e. 3 MFMAs per column tile a, 16 (+ 1 for the gamma p^T term) column tiles, and the result lands as
//   W[px = lane & 15][a][b = 4 g + r]: the contraction with F[px][a] (16 floats of the lane's own pixel, from the tile
//   image) is 64 FMAs per lane and half, and the lane ends with accF[px][4 g + r] of its group -- no cross-lane step.
// One wave therefore needs [y | Cinv'] (72 registers) AND the Z / p pieces (136) as static operands: 1 wave per SIMD,
// launch_bounds(256, 1), the 512-register file (VGPR + AGPR halves) of the SIMD to itself; MFMA and VALU work of
// DIFFERENT half-steps are independent instruction streams of the same wave:
//     step u:  phase A   MFMA  stage 3 of half u - 1   ||  VALU  stage 2 of half u
//              phase B   MFMA  stage 1 of half u + 1   ||  VALU  contraction of half u - 1, bf16 pieces of beta / gamma (u)
// One barrier per half-step (image ring hand-over and the four groups' partial sums); image halves arrive by LDS-DMA
// THREE half-steps ahead, spectra two tiles ahead (the staging scheme of k_grads_x).
#pragma once
#include "qfa_common.h"
#include "qfa_xdl_kernels.h"
#include "qfa_grads_x.h"      // split8, SpecA, f32x2

template <int KP_>
struct GWT {
    static constexpr int KP = KP_, KK2 = KP * (KP + 1) / 2;
    static constexpr int NKS = 1 + (KK2 + 31) / 32;      // K-steps of stage 1: [y, 0 | pair products, 32 per step]: 6 / 3
    static constexpr int S1_HALF = NKS * 3 * 1024;       // stage-1 image of one 16-pixel half: [K-step][piece][lane][8 k] bf16
    static constexpr int FROW = KP + 4;                  // floats per pixel row of the F block (pad: conflict-free b128 reads)
    static constexpr int AUX_F = 0;                      // aux block: F[16 px][FROW] float32, then the per-pixel parameters
    static constexpr int AUX_PAR = 16 * FROW * 4;        // Psi[16] | omega[16] | ti[16] | pwi[16] | l2i[16] (float32)
    static constexpr int AUX_B = 2048;
    static constexpr int HALF_B = S1_HALF + AUX_B;       // bytes of a half in global memory
    static constexpr int TILE_B = 2 * HALF_B;
    static constexpr int NCH_IMG = S1_HALF / 1024;       // 1-KiB DMA pieces of the stage-1 image of a half (18 / 9)
    static constexpr int NCH = NCH_IMG + AUX_B / 1024;   // + 2 for the aux block
    static constexpr int NG = 4, SPB = 64;
    static constexpr int RING_IMG = 3, RING_AUX = 5;
    static constexpr int STG_ARR = 16 * 128;             // staging: one array of one tile, [16 rows][32 px] float
    static constexpr int STG_MASK = 3 * STG_ARR;
    static constexpr int STG_B = 3 * STG_ARR + 512;
    static constexpr int PROW = KP + 4;                  // floats per pixel row of a partial-sum slot
    static constexpr int PARTF = 32 * PROW;              // floats of one group's F partial of a tile: [32 px][PROW]
    // LDS (bytes)
    static constexpr int L_IMG = 0;                                     // [RING_IMG][S1_HALF]
    static constexpr int L_AUX = L_IMG + RING_IMG * S1_HALF;            // [RING_AUX][AUX_B]
    static constexpr int L_PART = L_AUX + RING_AUX * AUX_B;             // [2 tile parity][NG][PARTF] float
    static constexpr int L_PSUM = L_PART + 2 * NG * PARTF * 4;          // [2][NG][4 sums][32 px] float
    static constexpr int L_STG = L_PSUM + 2 * NG * 512;                 // [NG waves][2 tile parity][STG_B]
    static constexpr int L_TOTAL = L_STG + NG * 2 * STG_B;
};
static_assert(GWT<16>::L_TOTAL <= 160 * 1024 && GWT<8>::L_TOTAL <= 160 * 1024, "k_grads_w LDS");
static_assert(GWT<16>::AUX_PAR + 5 * 64 <= GWT<16>::AUX_B, "aux block");

// ------------------------------------------------------------------------------------------------
// k_prep_pgw : F, Psi, omega (+ the per-pixel factors of the factored-z form) -> the pass-2 image, one block per 32-pixel tile
//   half h:  [K-step ks][piece][lane (g, lo)][8 k] bf16 : B[k = 32 ks + 8 g + j][px = 2 lo + h]   (as k_prep_pgx)
//            aux: F[lo][a] float32 (row stride FROW), Psi[lo], omega[lo], ti[lo], pwi[lo], l2i[lo] of the pixels 2 lo + h
// ZP != NULL (factored-z form): the per-pixel factors {ti, pwi, l2i} of qfa_common.h
// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void k_prep_pgw(qfa_params_t p, const float4 *__restrict__ ZP,
                                                  int Npix, int Nb, int Nh, unsigned char *__restrict__ PGW) {
    using GW = GWT<KP>;
    const float *__restrict__ F = p.F;
    unsigned char *tile = PGW + (size_t)blockIdx.x * GW::TILE_B;
    const int p0 = 32 * blockIdx.x;
    __shared__ float f[32][17];
    for (int i = threadIdx.x; i < 32 * 16; i += 256) {
        const int px = i >> 4, a = i & 15;
        f[px][a] = (p0 + px < Npix && a < Nh) ? F[(size_t)(p0 + px) * Nh + a] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * GW::NKS * 64; i += 256) {
        const int lane = i & 63, ks = (i >> 6) % GW::NKS, h = i / (64 * GW::NKS);
        const int lo = lane & 15, g = lane >> 4, px = 2 * lo + h;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * g + j;
            float x = 0.f;
            if (ks == 0) {
                if (kk < KP) x = f[px][kk];
            } else {
                const int q = 32 * (ks - 1) + kk;
                if (q < GW::KK2) {
                    int a = 0;
                    while (a + 1 < KP && pair_index(a + 1, a + 1, KP) <= q) ++a;
                    const int b = a + (q - pair_index(a, a, KP));
                    x = f[px][a] * f[px][b];
                }
            }
            v[j] = x;
        }
        u32x4 ph, pm, pl;
        split8(v, ph, pm, pl);
        unsigned char *dst = tile + h * GW::HALF_B + ks * 3072 + lane * 16;
        *reinterpret_cast<u32x4 *>(dst) = ph;
        *reinterpret_cast<u32x4 *>(dst + 1024) = pm;
        *reinterpret_cast<u32x4 *>(dst + 2048) = pl;
    }
    for (int i = threadIdx.x; i < 2 * (GW::AUX_B / 4); i += 256) {
        const int h = i / (GW::AUX_B / 4), j = i % (GW::AUX_B / 4);
        float *aux = reinterpret_cast<float *>(tile + h * GW::HALF_B + GW::S1_HALF);
        float v = 0.f;
        if (j < 16 * GW::FROW) {
            const int lo = j / GW::FROW, a = j % GW::FROW;
            if (a < KP) v = f[2 * lo + h][a];
        } else if (j < 16 * GW::FROW + 80) {
            const int q = (j - 16 * GW::FROW) >> 4, lo = (j - 16 * GW::FROW) & 15, px = p0 + 2 * lo + h;
            if (q == 0) v = px < Npix ? p.Psi[px] : 0.f;
            else if (q == 1) v = px < Nb ? p.omega[px] : 0.f;
            else if (ZP && px < Nb) {
                const float4 zq = ZP[px];
                v = q == 2 ? zq.x : (q == 3 ? zq.y : zq.z);
            }
        }
        aux[j] = v;
    }
}

// dynamic operand of stage 3: four float32 values (the lane's spectra 4 g + r) as the three K = 32 operands
//   {h | l}, {m | m}, {h | h}      (two piece slots of 4 bf16 each)
struct DynOp {
    u32x4 hl, mm, hh;
};
__device__ __forceinline__ DynOp dyn_split(const float (&x)[4]) {
    unsigned h01, m01, l01, h23, m23, l23;
    split2(x[0], x[1], h01, m01, l01);
    split2(x[2], x[3], h23, m23, l23);
    DynOp o;
    o.hl = u32x4{h01, h23, l01, l23};
    o.mm = u32x4{m01, m23, m01, m23};
    o.hh = u32x4{h01, h23, h01, h23};
    return o;
}

#ifndef QFA_GW_ABL
#define QFA_GW_ABL 0       // timing-only ablations (wrong results): 1 no spectra staging, 2 no flush, 4 no image DMA,
#endif                     // 8 no stage-3 MFMAs, 16 no stage-1 MFMAs, 32 no stage 2, 64 no contraction
#ifndef QFA_GW_SCHED
#define QFA_GW_SCHED 1     // 1: sched_group_barrier interleave of the MFMA and VALU streams of a phase
#endif

// ------------------------------------------------------------------------------------------------
// k_grads_w.  One work item = (block of 64 spectra, range of 32-pixel tiles); 256 threads, wave w = spectra 16 w .. 16 w + 15.
// slab != NULL: deterministic mode (see k_grads_x).
// ------------------------------------------------------------------------------------------------
template <int KP, bool HASA, bool ZF>
__global__ __launch_bounds__(256, 1) void k_grads_w(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau, int B, int Npix, int Nb,
                                                    int Nh, int ntiles, WorkPlan wp,
                                                    const unsigned char *__restrict__ PGW,
                                                    const float *__restrict__ SOL, const float4 *__restrict__ ZS,
                                                    float *__restrict__ accum, float *__restrict__ slab,
                                                    double *__restrict__ slabS, int slab_stride,
                                                    Scal64 *__restrict__ sc64) {
    using C = Cfg<KP>;
    using GW = GWT<KP>;
    constexpr int NA = KP;                                   // column tiles of stage 3 (one per a)
    __shared__ __attribute__((aligned(16))) unsigned char lds[GW::L_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = wave_uniform(tid >> 6);                    // group of 16 spectra inside the block
    const int lo = lane & 15, g = lane >> 4;
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int s0 = blk * GW::SPB + w * 16;
    const bool active = s0 < B;                              // wave-uniform
    const int n = t1 - t0;
    const int nbt = (Nb + 31) >> 5;                          // tiles that contain blue pixels
    const DevConsts k = load_consts(p, tau);

    const bool det = slab != nullptr;
    float *accF = det ? slab + (size_t)blk * (size_t)slab_stride : accum;
    float *accA = accF + (size_t)Npix * Nh;                  // sumA | gPsi | gOmega | cnt (contiguous)
    float *accS = accum + (size_t)Npix * Nh + 3 * (size_t)Npix + Nb;

    // zero the partial-sum slots (inactive groups never write theirs; row padding is never written)
    for (int i = tid; i < (GW::L_STG - GW::L_PART) / 4; i += 256) reinterpret_cast<float *>(lds + GW::L_PART)[i] = 0.f;

    const int rot = n > 0 ? (int)(((unsigned)blk * 2654435761u) % (unsigned)min(n, 32)) : 0;
    auto tile_of = [&](int c) {
        int x = c + rot;
        if (x >= n) x -= n;
        return t0 + x;
    };

    // ---------------------------------------------------------------- static operands
    // stage 1, A operand: spectrum s0 + lo, k = 32 ks + 8 g + j of [y, 0 | Cinv']
    u32x4 S1h[GW::NKS], S1m[GW::NKS], S1l[GW::NKS];
    {
        const bool v = active && (s0 + lo) < B;
        const float *sol = SOL + (size_t)(v ? s0 + lo : 0) * C::NSOL;
#pragma unroll
        for (int ks = 0; ks < GW::NKS; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = 8 * g + j;
                float val = 0.f;
                if (ks == 0) {
                    if (v && kk < KP) val = sol[kk];
                } else {
                    const int q = 32 * (ks - 1) + kk;
                    if (v && q < GW::KK2) val = sol[C::SOL_CI + q];
                }
                x[j] = val;
            }
            split8(x, S1h[ks], S1m[ks], S1l[ks]);
        }
    }
    // stage 3, A operands: row m = b = lo, k = 8 g + j <-> (spectrum 4 g + (j & 3), slot j >> 2):
    //   ZA1[a] = {Zl | Zh},  ZA2[a] = {Zh | Zm}  of Z_s[a][b];   PA1 = {pl | ph}, PA2 = {ph | pm} of p_s[b]
    u32x4 ZA1[NA], ZA2[NA], PA1, PA2;
    {
        const float *solr[4];
        bool vr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = s0 + 4 * g + r;
            vr[r] = active && s < B && lo < Nh;
            solr[r] = SOL + (size_t)(vr[r] ? s : 0) * C::NSOL;
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = (vr[r] && lo < KP) ? solr[r][C::SOL_Z + a * KP + (lo & (KP - 1))] : 0.f;
            unsigned h01, m01, l01, h23, m23, l23;
            split2(x[0], x[1], h01, m01, l01);
            split2(x[2], x[3], h23, m23, l23);
            ZA1[a] = u32x4{l01, l23, h01, h23};
            ZA2[a] = u32x4{h01, h23, m01, m23};
        }
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = (vr[r] && lo < KP) ? solr[r][C::SOL_P + (lo & (KP - 1))] : 0.f;
        unsigned h01, m01, l01, h23, m23, l23;
        split2(x[0], x[1], h01, m01, l01);
        split2(x[2], x[3], h23, m23, l23);
        PA1 = u32x4{l01, l23, h01, h23};
        PA2 = u32x4{h01, h23, m01, m23};
    }
    bool sv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) sv[r] = active && (s0 + 4 * g + r) < B;
    // factored-z form: the per-spectrum factors of the lane's four spectra
    float zs_ts[4], zs_pw[4], zs_l2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        zs_ts[r] = 0.f; zs_pw[r] = 0.f; zs_l2[r] = 0.f;
        if (ZF && sv[r]) {
            const float4 q = ZS[s0 + 4 * g + r];
            zs_ts[r] = q.x; zs_pw[r] = q.y; zs_l2[r] = q.z;
        }
    }
    const int last_row = active ? min(15, B - 1 - s0) : 0;   // wave-uniform
    const float *dbase = uniform_ptr(bt.delta + (size_t)(active ? s0 : 0) * Npix);
    const float *ebase = uniform_ptr(bt.error + (size_t)(active ? s0 : 0) * Npix);
    const uint8_t *mbase = uniform_ptr(bt.mask + (size_t)(active ? s0 : 0) * Npix);
    const float *zbase = ZF ? dbase : uniform_ptr(bt.zabs + (size_t)(active ? s0 : 0) * Nb);
    const float *abase = HASA ? bt.A_blue + (size_t)(active ? s0 : 0) * Nb : nullptr;
    double s_tau0 = 0.0, s_c0 = 0.0, s_beta = 0.0;           // float32 inside a half, float64 across

    // ---------------------------------------------------------------- spectra staging (as k_grads_x, role A)
    unsigned char *stg = lds + GW::L_STG + w * 2 * GW::STG_B;
    auto stage_tile = [&](int tg, int par) -> int {
        if (QFA_GW_ABL & 1) return 0;
        const bool zblue = !ZF && tg < nbt;                                               // wave-uniform
        const bool fastp = 32 * tg + 31 < Npix, fastz = ZF || !zblue || 32 * tg + 31 < Nb;
        const float *zb = zblue ? zbase : dbase;
        const int zlen = zblue ? Nb : Npix;
        const unsigned dst = wave_uniform(lds_addr(stg + par * GW::STG_B));
#if QFA_TRACKED_LOADS
        {
            float *sf = reinterpret_cast<float *>(stg + par * GW::STG_B);
            unsigned char *mb = stg + par * GW::STG_B + GW::STG_MASK;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = 2 * i + (lane >> 5), pxl = lane & 31;
                const unsigned row = (unsigned)min(q ^ ((q >> 2) & 1), last_row);
                const int px = 32 * tg + pxl;
                const unsigned o = row * (unsigned)Npix + (unsigned)min(px, Npix - 1);
                sf[0 * (GW::STG_ARR / 4) + q * 32 + pxl] = dbase[o];
                sf[1 * (GW::STG_ARR / 4) + q * 32 + pxl] = ebase[o];
                if (!ZF) sf[2 * (GW::STG_ARR / 4) + q * 32 + pxl] = zb[row * (unsigned)zlen + (unsigned)min(px, zlen - 1)];
                mb[q * 32 + pxl] = px < Npix ? mbase[o] : (unsigned char)0;
            }
            (void)dst; (void)fastz;
            return -1;
        }
#endif
        if (fastp) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = 8 * i + (lane >> 3);
                const unsigned row = (unsigned)min(q ^ ((q >> 2) & 1), last_row);
                const unsigned pc = 4u * (unsigned)(lane & 7);
                const unsigned o = row * (unsigned)Npix + 32u * (unsigned)tg + pc;
                glds16a(dbase, 4u * o, dst + 0 * GW::STG_ARR + i * 1024);
                glds16a(ebase, 4u * o, dst + 1 * GW::STG_ARR + i * 1024);
                if (!ZF && fastz) glds16a(zb, 4u * (row * (unsigned)zlen + 32u * (unsigned)tg + pc), dst + 2 * GW::STG_ARR + i * 1024);
                glds4a(mbase, o, dst + GW::STG_MASK + i * 256);
            }
            if (fastz) return ZF ? 6 : 8;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = 2 * i + (lane >> 5);
            const unsigned row = (unsigned)min(q ^ ((q >> 2) & 1), last_row);
            const int pxl = lane & 31;
            if (!ZF) glds4a(zb, 4u * (row * (unsigned)zlen + (unsigned)min(32 * tg + pxl, zlen - 1)), dst + 2 * GW::STG_ARR + i * 256);
            if (!fastp) {
                const unsigned o = row * (unsigned)Npix + (unsigned)min(32 * tg + pxl, Npix - 1);
                glds4a(dbase, 4u * o, dst + 0 * GW::STG_ARR + i * 256);
                glds4a(ebase, 4u * o, dst + 1 * GW::STG_ARR + i * 256);
            }
        }
        if (fastp) return 14;
        unsigned char *mb = stg + par * GW::STG_B + GW::STG_MASK;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = 2 * i + (lane >> 5);
            const unsigned row = (unsigned)min(q ^ ((q >> 2) & 1), last_row);
            const int px = 32 * tg + (lane & 31);
            mb[q * 32 + (lane & 31)] = px < Npix ? mbase[row * (unsigned)Npix + (unsigned)px] : (unsigned char)0;
        }
        return -1;                                            // ordinary loads among the requests: the next wait is for everything
    };
    auto take_tile = [&](int par, SpecA &cur) {
        const unsigned char *sb = stg + par * GW::STG_B + 8 * lo;
        const unsigned char *mb = stg + par * GW::STG_B + GW::STG_MASK + 2 * lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int slot = 4 * g + (r ^ (g & 1));
            cur.d[r] = *reinterpret_cast<const f32x2 *>(sb + 0 * GW::STG_ARR + slot * 128);
            f32x2 e2 = *reinterpret_cast<const f32x2 *>(sb + 1 * GW::STG_ARR + slot * 128);
            if (!ZF) cur.z[r] = *reinterpret_cast<const f32x2 *>(sb + 2 * GW::STG_ARR + slot * 128);
            const unsigned mk = *reinterpret_cast<const unsigned short *>(mb + slot * 32);
            e2[0] = (mk & 0xffu) ? fabsf(e2[0]) : -1.f;
            e2[1] = (mk & 0xff00u) ? fabsf(e2[1]) : -1.f;
            cur.sg[r] = e2;
        }
    };

    // ---------------------------------------------------------------- image DMA: half u -> image slot u % 3, aux slot u % 5
    auto get_half = [&](int u) -> int {
        if ((QFA_GW_ABL & 4) && u > 2) return 0;
        const int c = u >> 1, h = u & 1;
        const unsigned char *sbase = uniform_ptr(PGW + (size_t)tile_of(c) * GW::TILE_B + h * GW::HALF_B);
        unsigned char *img = lds + GW::L_IMG + (u % GW::RING_IMG) * GW::S1_HALF;
        unsigned char *aux = lds + GW::L_AUX + (u % GW::RING_AUX) * GW::AUX_B;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < (GW::NCH + GW::NG - 1) / GW::NG; ++i) {
            const int ch = w + GW::NG * i;
            if (ch < GW::NCH_IMG) {
                glds16a(sbase + ch * 1024, (unsigned)lane * 16u, wave_uniform(lds_addr(img + ch * 1024)));
                ++cnt;
            } else if (ch < GW::NCH) {
                glds16a(sbase + ch * 1024, (unsigned)lane * 16u, wave_uniform(lds_addr(aux + (ch - GW::NCH_IMG) * 1024)));
                ++cnt;
            }
        }
        return cnt;
    };

    // ---------------------------------------------------------------- flushes (one tile: 32 px x Nh sums of the four groups)
    const bool wide = det && (Nh & 3) == 0;
    constexpr int NWIDE = 8 * KP;                              // threads of the 16-byte form
    float *sink = accF + (slab_stride - 64) + lane;
    auto flush_F = [&](int tg, int par) -> int {
        if (QFA_GW_ABL & 2) return 0;
        const float *pp = reinterpret_cast<const float *>(lds + GW::L_PART + par * GW::NG * GW::PARTF * 4);
        if (wide) {
            if (tid >= NWIDE) return 0;                                                // wave-uniform
            const int pxl = tid / (KP / 4), b4 = 4 * (tid % (KP / 4));
            const int px = 32 * tg + pxl;
            const float4 *q4 = reinterpret_cast<const float4 *>(pp + pxl * GW::PROW + b4);
            const float4 v0 = q4[0], v1 = q4[GW::PARTF / 4], v2 = q4[2 * GW::PARTF / 4], v3 = q4[3 * GW::PARTF / 4];
            const float4 v = {(v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y),
                              (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w)};
            const bool ok = (b4 < Nh) & (px < Npix);
            if (ok) *reinterpret_cast<float4 *>(accF + (size_t)px * Nh + b4) = v;
            else *sink = v.x;
            return 1;
        }
#pragma unroll
        for (int k4 = 0; k4 < 32 * KP / 256; ++k4) {
            const int o = tid + 256 * k4;
            const int pxl = o / KP, bb = o % KP;
            const float *q = pp + pxl * GW::PROW + bb;
            const float v = (q[0] + q[GW::PARTF]) + (q[2 * GW::PARTF] + q[3 * GW::PARTF]);
            const int px = 32 * tg + pxl;
            const bool ok = (bb < Nh) & (px < Npix);
            if (det) *(ok ? accF + (size_t)px * Nh + bb : sink) = v;
            else atomicAdd(accF + (size_t)min(px, Npix - 1) * Nh + bb % Nh, ok ? v : 0.f);
        }
        return 32 * KP / 256;
    };
    // per-pixel sums [sumA | gPsi | gOmega | cnt] of tile tg: thread (which = t >> 5, pxl = t & 31) of the first 128 --
    // waves 0 and 1 (waves 2 and 3 when the F sums go out as 16-byte stores)
    auto flush_P = [&](int tg, int par) -> int {
        if (QFA_GW_ABL & 2) return 0;
        if (wide ? tid < 128 : tid >= 128) return 0;                                   // wave-uniform
        const int which = (tid >> 5) & 3, pxl = tid & 31;
        const float *q = reinterpret_cast<const float *>(lds + GW::L_PSUM + par * GW::NG * 512) + which * 32 + pxl;
        const float v = (q[0] + q[128]) + (q[256] + q[384]);
        const int px = 32 * tg + pxl;
        const bool ok = (px < Npix) & ((which != 2) | (px < Nb));
        const int pxc = min(px, Npix - 1);
        const int offc = (which == 2 && pxc >= Nb) ? 2 * Npix + Nb + pxc : which * Npix - (which == 3 ? Npix - Nb : 0) + pxc;
        if (det) *(ok ? accA + offc : sink) = v;
        else atomicAdd(accA + offc, ok ? v : 0.f);
        return 1;
    };

    // ---------------------------------------------------------------- the pipeline registers
    f32x4 afy = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f};        // stage-1 result of the half stage 2 works on next
    DynOp Bb, Bg;                                                       // beta / gamma operands of the half stage 3 works on next
    Bb.hl = Bb.mm = Bb.hh = Bg.hl = Bg.mm = Bg.hh = u32x4{0u, 0u, 0u, 0u};
    SpecA cur;
#pragma unroll
    for (int r = 0; r < 4; ++r) cur.d[r] = cur.sg[r] = cur.z[r] = f32x2{0.f, 0.f};

    // stage 1 of half u (image slot u % 3): 6 K-steps x 6 MFMAs; B pieces of K-step ks + 1 are read under the MFMAs of ks
    auto stage1 = [&](int u, f32x4 &ofy, f32x4 &oq) {
        const unsigned char *bp = lds + GW::L_IMG + (u % GW::RING_IMG) * GW::S1_HALF + lane * 16;
        f32x4 fy = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
        u32x4 bq[2][3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bq[0][pc] = *reinterpret_cast<const u32x4 *>(bp + pc * 1024);
#pragma unroll
        for (int ks = 0; ks < GW::NKS; ++ks) {
            if (ks + 1 < GW::NKS) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    bq[(ks + 1) & 1][pc] = *reinterpret_cast<const u32x4 *>(bp + (ks + 1) * 3072 + pc * 1024);
            }
            const u32x4 &bh = bq[ks & 1][0], &bm = bq[ks & 1][1], &bl = bq[ks & 1][2];
            if (!(QFA_GW_ABL & 16)) {
                if (ks == 0) fy = xdl6(S1h[ks], S1m[ks], S1l[ks], bh, bm, bl, fy);
                else q = xdl6(S1h[ks], S1m[ks], S1l[ks], bh, bm, bl, q);
            }
        }
        ofy = fy;
        oq = q;
    };

    // stage 3 of a half: W^T tiles from the beta operand, the gamma p^T term; then (later) the contraction with F
    auto stage3 = [&](const DynOp &ob, const DynOp &og, f32x4 (&W)[NA], f32x4 &accP) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (QFA_GW_ABL & 8) {
#pragma unroll
            for (int a = 0; a < NA; ++a) W[a] = zero;
            accP = zero;
            return;
        }
        accP = xdl(PA2, og.hh, xdl(PA2, og.mm, xdl(PA1, og.hl, zero)));
#pragma unroll
        for (int a = 0; a < NA; ++a) W[a] = xdl(ZA2[a], ob.hh, xdl(ZA2[a], ob.mm, xdl(ZA1[a], ob.hl, zero)));
    };

    // stage 2 of half u = (tile c, half h): the lane's four elements (pixel 2 lo + h of the spectra 4 g + r)
    float t_tau0 = 0.f, t_c0 = 0.f, t_beta = 0.f;
    auto stage2 = [&](auto blue_tag, int tg, int h, int u, const f32x4 &fy, const f32x4 &q, float (&betaR)[4], float (&gamR)[4]) {
        constexpr bool BLUE = decltype(blue_tag)::value;
        const float *par = reinterpret_cast<const float *>(lds + GW::L_AUX + (u % GW::RING_AUX) * GW::AUX_B + GW::AUX_PAR);
        float *psum = reinterpret_cast<float *>(lds + GW::L_PSUM + (((u >> 1) & 1) * GW::NG + w) * 512);
        const float Psi = par[lo], om = par[16 + lo];
        float ti = 0.f, pwi = 0.f, l2i = 0.f;
        if (ZF && BLUE) { ti = par[32 + lo]; pwi = par[48 + lo]; l2i = par[64 + lo]; }
        const int px = 32 * tg + 2 * lo + h;
        const bool inb = px < Npix;
        const bool blue = px < Nb;
        float gPsi = 0.f, gOm = 0.f, sA = 0.f, cnt = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool wv_ = inb & sv[r] & (__float_as_int(cur.sg[r][h]) >= 0);
            const float dd = wv_ ? cur.d[r][h] : 0.f;
            const float sg = cur.sg[r][h];
            if (BLUE) {
                float l2, pw, Ab;
                if (ZF) {
                    l2 = zs_l2[r] + l2i;
                    pw = zs_pw[r] * pwi;
                    Ab = fast_exp2(fmaf(zs_ts[r], ti, k.offp));                                   // QFA/model.py:125
                } else {
                    l2 = fast_log2(1.0f + cur.z[r][h]);
                    pw = fast_exp2(k.beta * l2);
                    const float tauv = k.t_amp * fast_exp2(k.t_expo * (l2 + k.t_lscale)) + k.t_off;   // QFA/utils.py:105-141
                    Ab = fast_exp2(-tauv * QFA_LOG2E);
                }
                if (HASA) Ab = abase[(unsigned)min(4 * g + r, last_row) * (unsigned)Nb + (unsigned)min(px, Nb - 1)];
                const float re = k.omc0 - fast_exp2(k.k1 * pw);                                   // QFA/utils.py:91
                const float Av = blue ? Ab : 1.f;
                const float zd = blue ? re * re : 0.f;
                const float A2 = Av * Av;
                const float D = A2 * Psi + om * zd + sg * sg;
                const float wD = wv_ ? fast_rcp(D) : 0.f;
                const float wDA = wD * Av;
                const float uu = wD * (dd - Av * fy[r]);                    // (Sigma^-1 delta)_i
                const float dS = wD - wDA * wDA * q[r];                     // diag(Sigma^-1)_i
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
                const float uu = wD * (dd - fy[r]);
                const float dS = wD - wD * wD * q[r];
                gPsi += 0.5f * (dS - uu * uu);
                cnt += wv_ ? 1.f : 0.f;
                betaR[r] = wD;
                sA += wD;
                gamR[r] = uu;
            }
        }
        // per-pixel sums over the wave's 16 spectra (lanes lo + 16 g'): quantity g's sum ends in row g
        {
            const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sA), __float_as_uint(gPsi), false, false);
            const auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(gOm), __float_as_uint(cnt), false, false);
            const float u01 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);
            const float u23 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);
            const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(u01), __float_as_uint(u23), false, false);
            psum[g * 32 + 2 * lo + h] = __uint_as_float(t[0]) + __uint_as_float(t[1]);
        }
        if (BLUE) {
            s_tau0 += (double)t_tau0;
            s_c0 += (double)t_c0;
            s_beta += (double)t_beta;
            t_tau0 = 0.f; t_c0 = 0.f; t_beta = 0.f;
        }
    };

    // contraction of half v = (tile cv, half hv): accF[px][4 g + r] = accP[r] + sum_a F[px][a] W[a][r] -> the group's slot
    auto contract = [&](int v, const f32x4 (&W)[NA], const f32x4 &accP) {
        const float *fp = reinterpret_cast<const float *>(lds + GW::L_AUX + (v % GW::RING_AUX) * GW::AUX_B + GW::AUX_F) + lo * GW::FROW;
        float *part = reinterpret_cast<float *>(lds + GW::L_PART + ((((v >> 1) & 1) * GW::NG + w) * GW::PARTF) * 4);
        f32x4 acc = accP;
        if (!(QFA_GW_ABL & 64)) {
#pragma unroll
            for (int a4 = 0; a4 < NA / 4; ++a4) {
                const float4 f4 = *reinterpret_cast<const float4 *>(fp + 4 * a4);
                const float fa[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = fmaf(fa[j], W[4 * a4 + j][r], acc[r]);
                }
            }
        }
        const int pxl = 2 * lo + (v & 1);
        if (KP == 16 || g < KP / 4)
            *reinterpret_cast<float4 *>(part + pxl * GW::PROW + 4 * g) = float4{acc[0], acc[1], acc[2], acc[3]};
    };

    // ---------------------------------------------------------------- prologue: image halves 0, 1, 2 and the first two tiles' spectra
    const int U = 2 * n;                                     // half-steps with data
    int q_prev = 0;                                          // requests this wave issued in the previous step (-1: wait for all)
    {
        if (U > 0) get_half(0);
        if (U > 1) get_half(1);
        if (n > 0 && active) {
            stage_tile(tile_of(0), 0);
            if (n > 1) stage_tile(tile_of(1), 1);
        }
        dma_wait<0>();
        step_barrier();
        if (U > 2) q_prev = get_half(2);
        if (active && U > 0 && !(QFA_GW_ABL & 16)) stage1(0, afy, aq);      // (step u = -1, phase B)
    }

    // ---------------------------------------------------------------- the loop over half-steps u = 0 .. U + 1
    auto step = [&](auto blue_tag, int u) {
        const int c = u >> 1, h = u & 1;
        // everything issued before the previous step has landed (image u + 1, the spectra of tile c), LDS hand-over
        if (q_prev < 0) dma_wait<0>();
        else dma_wait_n(q_prev);
        step_barrier();
        int q = 0;
        if (u + 3 < U) q += get_half(u + 3);
        const bool work = u < U && active;
        const int tg = u < U ? tile_of(c) : 0;
        int qs = 0;
        if (work && h == 0) {
            take_tile(c & 1, cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // staging buffer read: it may be overwritten now
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < n) qs = stage_tile(tile_of(c + 2), c & 1);
        }
        // the flush of tile c - 1 (its last contraction was in phase B of step 2 c = u - 1)
        int qf = 0;
        if (h == 1 && c >= 1) {
            qf += flush_F(tile_of(c - 1), (c - 1) & 1);
            qf += flush_P(tile_of(c - 1), (c - 1) & 1);
        }
        q_prev = qs < 0 ? -1 : q + qs + qf;

        // ---- phase A: MFMA stage 3 of half u - 1  ||  VALU stage 2 of half u
        f32x4 W[NA], accP;
        float betaR[4] = {0.f, 0.f, 0.f, 0.f}, gamR[4] = {0.f, 0.f, 0.f, 0.f};
        const bool have3 = active && u >= 1 && u <= U;
        if (have3) stage3(Bb, Bg, W, accP);
        if (work && !(QFA_GW_ABL & 32)) stage2(blue_tag, tg, h, u, afy, aq, betaR, gamR);
        // ---- phase B: MFMA stage 1 of half u + 1  ||  VALU contraction of half u - 1, pieces of beta / gamma (u)
        if (active && u + 1 < U) stage1(u + 1, afy, aq);
        if (have3) contract(u - 1, W, accP);
        if (work) {
            Bb = dyn_split(betaR);
            Bg = dyn_split(gamR);
        }
    };
    for (int c = 0; c <= n; ++c) {
        const bool blue_tile = c < n && tile_of(c) < nbt;
        if (blue_tile) {
            step(std::true_type{}, 2 * c);
            step(std::true_type{}, 2 * c + 1);
        } else {
            step(std::false_type{}, 2 * c);
            step(std::false_type{}, 2 * c + 1);
        }
    }
    dma_wait<0>();

    // ---------------------------------------------------------------- scalar gradients
    if (active) {
        for (int o = 32; o >= 1; o >>= 1) {
            s_tau0 += __shfl_xor(s_tau0, o);
            s_c0 += __shfl_xor(s_c0, o);
            s_beta += __shfl_xor(s_beta, o);
        }
    }
    if (det) {
        if (lane == 0) {
            double *q = slabS + ((size_t)blockIdx.x * GW::NG + w) * 3;
            q[0] = active ? s_tau0 : 0.0; q[1] = active ? s_c0 : 0.0; q[2] = active ? s_beta : 0.0;
        }
    } else {
        scal64_commit(sc64, active ? s_tau0 : 0.0, active ? s_c0 : 0.0, active ? s_beta : 0.0, gridDim.x * (unsigned)GW::NG, accS);
    }
}
