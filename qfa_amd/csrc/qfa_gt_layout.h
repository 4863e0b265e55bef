// qfa_gt_layout.h -- the pixel-resident form of pass 2 (qfa_grads_t.h): sizes of its images and LDS rings, and the builder of
// the per-group operand images ("state"), which runs at the end of k_solve (qfa_step_kernels.h) on the solve's output
// while it is still in LDS -- or as a kernel of its own (k_prep_pst) on the float32 record SOL.
#pragma once
#include "qfa_common.h"

#ifndef QFA_GT_F16S1
#define QFA_GT_F16S1 1     // stage 1 of k_grads_t on TWO float16 pieces per operand and three products (18 MFMAs per group instead of 36;
                           // qfa_common.h "float16 pieces"): the image carries a power of two per pixel, the state one per spectrum
#endif
#ifndef QFA_GT_F16S3
#define QFA_GT_F16S3 1     // stage 3 of k_grads_t: W += Z beta on TWO float16 pieces per operand, three products in two MFMAs per column tile
                           // (35 MFMAs per group instead of 51).  beta = A^2 / D <= 1 / Psi of its PIXEL whatever the data (D >= A^2 Psi): the
                           // image carries the power of two that brings beta below 2^12 and W is a per-pixel sum -- the scale leaves once, when F
                           // enters.  Z of a spectrum is scaled by 2^7 (by less where its largest element exceeds 128; the factor rides with the
                           // spectrum's stage-1 scales and multiplies its beta)
#endif
template <int KP_>
struct GTT {
    static constexpr int KP = KP_, KK2 = KP * (KP + 1) / 2;
    // K axis of stage 1: the KK2 pair products, then (in the free slots of the last 32-wide block, from slot YOFF) the KP
    // values of F / y.  The pixel side (B operand, in registers for the whole walk) is ONE image of NKQ blocks; the spectrum
    // side has NKQ blocks [Cinv' | 0] for f^T Cinv f and one more block [0 | y | 0] that meets the last image block again
    // for f^T y: 36 MFMAs as with a separate y block, 60 instead of 72 registers.
    static constexpr int NKQ = (KK2 + 31) / 32;              // 5
    static constexpr int YOFF = (KK2 % 32 + 7) / 8 * 8;      // 8
    static_assert(KK2 % 32 != 0 && YOFF + KP <= 32, "F / y share the last pair block");
    static constexpr int NKS = NKQ + 1;                      // blocks of the spectrum side (6)
    static constexpr bool F16S1 = QFA_GT_F16S1 != 0;
    static constexpr int S1NP = F16S1 ? 2 : 3;               // pieces per K block of stage 1 (float16 h, m | bf16 h, m, l)
    static constexpr int BLK_B = S1NP * 1024;
    // per 16-pixel tile in global memory (k_prep_pgt)
    static constexpr int IMG_B = NKQ * BLK_B;                // [ks][piece][lane (g, lo)][8 k]: B[k = 32 ks + 8 g + j][px = lo]
    // float Psi[16] | omega[16] | ti[16] | pwi[16] | l2i[16] | F16S1: 1 / t^2 [16] | 1 / t [16] (t = the pixel's power of two:
    // the image holds t^2 f_a f_b and t f_a)
    static constexpr int OFF_PAR = IMG_B;
    static constexpr int PAR_IT2 = 80, PAR_IT1 = 96, PAR_SBETA = 112;      // float index in the parameter block: 1 / t^2, 1 / t, s_beta of pixel lo
    static constexpr int OFF_F = IMG_B + 512;                // float F[16 px][KP]
    static constexpr int TILE_B = (OFF_F + 16 * KP * 4 + 1023) / 1024 * 1024;
    // per group of 16 spectra in global memory (k_prep_pst)
    // F16S1: Cinv' and y of a spectrum are scaled by powers of two of their own; their inverses sit as float32 in the h piece of
    // the y block, lanes g = 3 (K slots 24..31, which meet zeros of the image): float 2 s = 1 / scale(Cinv'), 2 s + 1 = 1 / scale(y)
    static constexpr int S1_B = NKS * BLK_B;                 // [block][piece][lane (g, lo = spectrum)][8 k]: A[s][k] of stage 1
    static constexpr int S1_SCALES = NKQ * BLK_B + 48 * 16;  // (byte offset of those 32 floats in the S1 part)
    static constexpr int S1P_B = S1_B;                       // (the S1 part as the ring holds it)
    // stage 3: column tiles of W.  KP = 16: one per a (rows m = b).  KP = 8: the 8 rows b of TWO a fill one 16-row MFMA tile
    // (row m <-> a = 2 tile + (m >> 3), b = m & 7): half the MFMAs, operand reads and accumulator registers
    static constexpr int APT = 16 / KP;                      // a per tile (1 or 2)
    static constexpr int NWT = KP / APT;                     // tiles (16 or 4)
    static constexpr bool F16S3 = QFA_GT_F16S3 != 0;
    static_assert(!F16S3 || F16S1, "the per-spectrum factor of stage 3 lives beside the stage-1 scales");
    static constexpr int ZT_B = F16S3 ? 1024 : 2048;         // bytes per column tile of the Z part
    static constexpr int Z_B = NWT * ZT_B;                   // bf16: [tile][operand 1, 2][lane (g, lo = row m)][4 dwords]; float16: [tile][lane][h01 h23 m01 m23]
    static constexpr int ZP_B = Z_B + 2 * 1024;              // + the p operands (gamma term: bf16 pieces, six products)
    static constexpr int S1_ZFAC = S1_SCALES + 128;          // F16S3: float32 2^(7 - zk) of the 16 spectra (Z is stored as Z 2^zk)
    static constexpr int STATE_B = S1P_B + ZP_B;
    static constexpr int NW = 8;                             // waves per workgroup
    // A wave owns TPW 16-pixel tiles = PXW pixels; tile j holds the pixels PXW wt + TPW lo + j (lo = the tile's column): a
    // lane's TPW pixels are adjacent, so are the bytes it reads from the staging buffer.  KP = 8: two tiles per wave -- the
    // image and W of a tile are 24 + 16 registers there, and every operand read, state piece and barrier serves two tiles
    static constexpr int TPW = KP == 8 ? 2 : 1;
    static constexpr int PXW = 16 * TPW;
    // a part moves as 1-KiB pieces, contiguous runs of them per wave (k_grads_t decides which waves)
    static constexpr int S1_PCS = S1P_B / 1024, Z_PCS = ZP_B / 1024;     // 12 (bf16 pieces: 18), 18 (34) (KP = 8: 6 (9), 6 (10))
    static_assert((Z_PCS + 4) / 5 <= 7 && (S1_PCS + 4) / 5 <= 6, "pieces per wave and stage: 7 slots in stage 2, 7+ in stage 3, NKS + 1 = 7 in stage 1 (4 at KP = 8)");
    // per-wave staging of the spectra of one group: [16 slots][16 px] float x 3 (delta, sigma, zabs -- or, factored-z form,
    // the float4 factors ZS of the 16 spectra), then mask bytes [16 slots][16]
    // (TPW = 2: [16 slots][32 px] float x 3, mask bytes as two halves [2][16 slots][16])
    // ABI v3 (rows): then the row indices of the 16 spectra of the group TWO groups later -- the request that stages group t
    // carries them, the lanes read them when they form the addresses of group t + 2 (k_grads_t, stage_spectra)
    static constexpr int STG_ARR = 1024 * TPW, STG_MASK = 3 * STG_ARR, STG_ROWS = STG_MASK + 256 * TPW, STG_B = STG_ROWS + 64;
    static constexpr int L_S1 = 0;                           // [2][S1P_B]
    static constexpr int L_Z = L_S1 + 2 * S1P_B;             // [2][ZP_B]
    static constexpr int L_STG = L_Z + 2 * ZP_B;             // [NW][2][STG_B]
    static constexpr int L_TOTAL = L_STG + NW * 2 * STG_B;
};
static_assert(GTT<16>::L_TOTAL <= 160 * 1024 && GTT<8>::L_TOTAL <= 160 * 1024, "k_grads_t LDS");

// ------------------------------------------------------------------------------------------------
// build_state : the solve's output of one group of 16 spectra (`rows`: 16 records of Cfg<KP>::NSOL floats, global memory or
// LDS) -> the group's state as split-bf16 MFMA operands, in the order the walk streams them.  256 threads.
//   S1 part: A[m = spectrum lo][k = 8 g + j of block ks] = ks < NKQ: Cinv'[pair 32 ks + 8 g + j] | ks = NKQ: y[8 g + j - YOFF]
//   Z  part: per column tile, operands ZA1 = {l01, l23, h01, h23}, ZA2 = {h01, h23, m01, m23} of
//            x[r] = Z_{4 g + r}[a][b] (row m = lo <-> (a, b): GTT::APT; k = 8 g + j <-> spectrum 4 g + (j & 3), piece slot j >> 2);
//            then the same two operands of p_{4 g + r}[b] (rows b < KP)
// ------------------------------------------------------------------------------------------------
template <int KP>
__device__ __forceinline__ void build_state(const float *rows, int s0, int B, int Nh, unsigned char *__restrict__ st, int tid) {
    using C = Cfg<KP>;
    using GT = GTT<KP>;
    // F16S1: the powers of two of the group's 16 spectra (16 threads per spectrum look at its Cinv' and y)
    __shared__ float sc_[16][6];                                         // scale(Cinv'), scale(y), their inverses; F16S3: 2^zk of Z, 2^(7 - zk)
    if constexpr (GT::F16S1) {
        __syncthreads();                                                 // (k_solve calls this per group: the previous call's readers)
        const int s = tid >> 4, sub = tid & 15;
        const bool v = s0 + s < B;
        const float *sol = rows + (size_t)(v ? s : 0) * C::NSOL;
        float mc = 0.f, my = 0.f, mz = 0.f;
        for (int q = sub; q < GT::KK2; q += 16) mc = fmaxf(mc, fabsf(sol[C::SOL_CI + q]));
        if (sub < KP) my = fabsf(sol[sub]);
        if (GT::F16S3) for (int q = sub; q < KP * KP; q += 16) mz = fmaxf(mz, fabsf(sol[C::SOL_Z + q]));
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            mc = fmaxf(mc, __shfl_xor(mc, o)); my = fmaxf(my, __shfl_xor(my, o)); mz = fmaxf(mz, __shfl_xor(mz, o));
        }
        if (sub == 0) {
            float ic, iy;
            const float c_ = f16_row_scale(v ? mc : 0.f, ic), y_ = f16_row_scale(v ? my : 0.f, iy);
            sc_[s][0] = c_; sc_[s][1] = y_; sc_[s][2] = ic; sc_[s][3] = iy;
            // Z 2^zk with zk = 7 while |Z| < 128 (Z = C^-1 T: of order one), less beyond (|Z 2^zk| < 2^14 always)
            int e = 0;
            if (v && mz > 0.f && mz < 3.0e38f) (void)frexpf(mz, &e);                  // mz = m 2^e, m in [0.5, 1)
            const int zk = e > 7 ? (e > 100 ? -86 : 14 - e) : 7;
            sc_[s][4] = ldexpf(1.f, zk); sc_[s][5] = ldexpf(1.f, 7 - zk);
        }
        __syncthreads();
    }
    for (int i = tid; i < GT::NKS * 64; i += 256) {
        const int lane = i & 63, ks = i >> 6, lo = lane & 15, g = lane >> 4;
        const bool v = s0 + lo < B;
        const float *sol = rows + (size_t)(v ? lo : 0) * C::NSOL;
        const float scl = GT::F16S1 ? sc_[lo][ks < GT::NKQ ? 0 : 1] : 1.f;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float val = 0.f;
            if (ks < GT::NKQ) {                                          // blocks 0 .. NKQ - 1: Cinv' (doubled off-diagonals)
                const int q = 32 * ks + 8 * g + j;
                if (v && q < GT::KK2) val = sol[C::SOL_CI + q];
            } else {                                                     // block NKQ: y in the slots of F in the last image block
                const int a = 8 * g + j - GT::YOFF;
                if (v && a >= 0 && a < KP) val = sol[a];
            }
            x[j] = val * scl;
        }
        unsigned char *dst = st + ks * GT::BLK_B + lane * 16;
        if constexpr (GT::F16S1) {
            u32x4 h, m;
            split8h(x, h, m);
            if (ks == GT::NKQ && g == 3) {                               // (zeros so far: the inverse scales of spectra 2 lo, 2 lo + 1)
                static_assert(GT::YOFF + KP <= 24, "K slots 24..31 of the y block are free");
                if (lo < 8) h = u32x4{__float_as_uint(sc_[2 * lo][2]), __float_as_uint(sc_[2 * lo][3]),
                                      __float_as_uint(sc_[2 * lo + 1][2]), __float_as_uint(sc_[2 * lo + 1][3])};
                else if (GT::F16S3 && lo < 12) h = u32x4{__float_as_uint(sc_[4 * (lo - 8)][5]), __float_as_uint(sc_[4 * (lo - 8) + 1][5]),
                                                         __float_as_uint(sc_[4 * (lo - 8) + 2][5]), __float_as_uint(sc_[4 * (lo - 8) + 3][5])};
            }
            *reinterpret_cast<u32x4 *>(dst) = h;
            *reinterpret_cast<u32x4 *>(dst + 1024) = m;
        } else {
            u32x4 h, m, l;
            split8(x, h, m, l);
            *reinterpret_cast<u32x4 *>(dst) = h;
            *reinterpret_cast<u32x4 *>(dst + 1024) = m;
            *reinterpret_cast<u32x4 *>(dst + 2048) = l;
        }
    }
    for (int i = tid; i < (GT::NWT + 1) * 64; i += 256) {
        const int lane = i & 63, wt = i >> 6, lo = lane & 15, g = lane >> 4;      // wt == NWT: the p operands
        const int a = wt * GT::APT + (lo >> (KP == 8 ? 3 : 4)), bcol = lo & (KP - 1);           // row lo of tile wt <-> Z[a][bcol]
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool v = s0 + 4 * g + r < B && bcol < Nh && (wt < GT::NWT || lo < KP);
            const float *sol = rows + (size_t)(v ? 4 * g + r : 0) * C::NSOL;
            x[r] = v ? (wt < GT::NWT ? sol[C::SOL_Z + a * KP + bcol] : sol[C::SOL_P + bcol]) : 0.f;
            if (GT::F16S3 && wt < GT::NWT) x[r] *= sc_[4 * g + r][4];
        }
        if (GT::F16S3 && wt < GT::NWT) {
            unsigned h01, m01, h23, m23;
            split2h(x[0], x[1], h01, m01);
            split2h(x[2], x[3], h23, m23);
            *reinterpret_cast<u32x4 *>(st + GT::S1P_B + wt * GT::ZT_B + lane * 16) = u32x4{h01, h23, m01, m23};
        } else {
            unsigned h01, m01, l01, h23, m23, l23;
            split2(x[0], x[1], h01, m01, l01);
            split2(x[2], x[3], h23, m23, l23);
            unsigned char *dst = st + GT::S1P_B + (wt < GT::NWT ? wt * GT::ZT_B : GT::Z_B) + lane * 16;
            *reinterpret_cast<u32x4 *>(dst) = u32x4{l01, l23, h01, h23};
            *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{h01, h23, m01, m23};
        }
    }
}
