// qfa_common.h -- shared device helpers and compile-time layout of the QFA hot path (gfx950).
//
// Notation: per spectrum s and pixel i (blue side = first Nb pixels)
//   A   = exp(-tau(zabs))            (1 on the red side)            reference QFA/model.py:125
//   zd  = (1 - c0 - exp(-tau0 (1+z)^beta))^2  (0 on the red side)   QFA/utils.py:91-92
//   D   = A^2 Psi + omega zd + sigma^2                              QFA/model.py:128-131
//   wD  = mask / D
//   C   = I + sum_i wD A^2 f_i f_i^T,  T = sum_i wD A^3 f_i f_i^T   (k x k, symmetric)
//   b   = sum_i wD A delta f_i,        b2 = sum_i wD A^2 delta f_i
// The k(k+1)/2 distinct products f_a f_b of every pixel are tabulated once per step in the
// "PF image" (row i = [f_i | P_i], P_i[pair(a,b)] = f_ia f_ib), so that C/T/b/b2 of 16 spectra
// are plain GEMMs against it and map onto v_mfma_f32_16x16x4_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/qfa_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define QFA_LOG2PI 1.8378770664093453f
#define QFA_LOG2E 1.4426950408889634f
#define QFA_LN2 0.6931471805599453f

template <int KP>
struct Cfg {
    static constexpr int FW = KP < 16 ? 16 : KP;       // width of the F part of a PF row
    static constexpr int KK2 = KP * (KP + 1) / 2;       // distinct pairs (a <= b)
    static constexpr int NT = (KK2 + 15) / 16;          // 16-column tiles of the pair part
    static constexpr int PW = NT * 16;                  // padded pair width
    static constexpr int NFT = FW / 16;                 // 16-column tiles of the F part
    // PF image (pass 1): row i = [f_i (FW) | f_ia f_ib (PW) | Psi_i, omega_i, 0, 0].  The stride is
    // 4 mod 8 floats so that the B-operand read of lane (j, col) at row 4j+e hits 32 distinct
    // LDS banks per half-wave (4*NCPL = 16 mod 32).
    static constexpr int PF_PSI = FW + PW;
    static constexpr int NCPL = FW + PW + 4;
    static constexpr int TILE_PF = 16 * NCPL;           // floats per 16-pixel tile (contiguous)
    // PFT image (pass 2), tile-major: tile t = [row c][16 px]; rows [0,KP) = F^T, [KP,KP+KK2) =
    // pair products, then Psi, omega, the per-pixel factors ti, pwi, l2i of the factored-z input form (ZP of this
    // header; zeros otherwise), zero padding to a multiple of 4 rows.
    static constexpr int PFT_PSI = KP + KK2;
    static constexpr int NR = (KP + KK2 + 5 + 3) / 4 * 4;    // (+ ti, pwi, l2i of the factored-z form behind Psi, omega)
    // N_h > 8: stage 3 of k_grads runs on the XDL pipe; its A operand (F of the tile as three bf16 pieces in the lane
    // order of the MFMA: KP = 16 v_mfma_f32_16x16x16_bf16 [piece][g][px][a = 4g + j], 3 x 512 bytes; KP = 32
    // v_mfma_f32_16x16x32_bf16 [piece][g][px][a = 8g + j], 3 x 1 KiB) is appended to the tile, and both parts are
    // padded to whole KiB so that the tile moves as 1-KiB LDS-DMA pieces
#ifndef QFA_XS3_32
#define QFA_XS3_32 1
#endif
    static constexpr bool XS3 = KP == 16 || (KP == 32 && QFA_XS3_32);
    static constexpr int PFT_MAIN = XS3 ? (NR * 16 + 255) / 256 * 256 : NR * 16;   // floats of the float32 part
#ifndef QFA_S3_F16
#define QFA_S3_F16 1       // k_grads_s3 (N_h = 17..32): F Z_s on two float16 pieces per operand, three products per spectrum instead of six
#endif
    // (KP = 32, QFA_S3_F16: behind the three bf16 pieces the same F as two float16 pieces of t_px F -- t_px the pixel's power of
    // two -- and a KiB whose first 16 floats are 1 / t_px: floats 768.., 1024.., 1280..)
    static constexpr int PFT_FP = XS3 ? (KP == 32 ? (QFA_S3_F16 ? 1536 : 768) : 512) : 0;                // floats
    static constexpr int PFT_F16H = 768, PFT_F16M = 1024, PFT_F16IT = 1280;      // (floats from PFT_MAIN: the float16 pieces and 1 / t)
    static constexpr int TILE_PFT = PFT_MAIN + PFT_FP;
    // per-spectrum moment record: [C PW][T PW][b FW][b2 FW][qd, ld, n, nblue]
    static constexpr int NMOM = 2 * PW + 2 * FW + 4;
    static constexpr int MOM_T = PW, MOM_B = 2 * PW, MOM_B2 = 2 * PW + FW, MOM_S = 2 * PW + 2 * FW;
    // per-spectrum solve record: [y KP][Cinv' KK2 (off-diagonals doubled)][Z KP*KP][p KP]
    static constexpr int NSOL = 2 * KP + KK2 + KP * KP;
    static constexpr int SOL_CI = KP, SOL_Z = KP + KK2, SOL_P = KP + KK2 + KP * KP;
};

// packed index of the pair (a, b), a <= b < KP: row-major upper triangle
__host__ __device__ constexpr int pair_index(int a, int b, int KP) {
    return a * KP - (a * (a - 1)) / 2 + (b - a);
}

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Work items of the two passes: one item = (block of 64 spectra, range of pixel tiles).  The first `full` blocks
// walk the whole pixel axis (they fill whole rounds of the resident-workgroup slots, one prologue each); the
// remaining `rem` blocks are cut into `nseg` pixel segments so that the last round is short (tail quantisation)
// and small batches use every CU.  One launch, items in this order.
#ifndef QFA_PX_SPW
#define QFA_PX_SPW 2      // posterior writer at N_h = 9..16: groups of 16 spectra per wave (k_predict_x; the plan's blocks are 64 SPW spectra)
                          // 2: every B-operand read from LDS feeds two MFMA chains -- 1.04 -> 0.90 ms at c3 (same box); at N_h <= 8
                          // (18 MFMAs per half) the same change costs 13 %
#endif
#ifndef QFA_PX_SPW8
#define QFA_PX_SPW8 1     // the same at N_h <= 8
#endif
struct WorkPlan {
    int full, rem, nseg, seg_tiles;
    __host__ __device__ int items() const { return full + rem * nseg; }
};
__device__ __forceinline__ void plan_item(const WorkPlan &w, int item, int ntiles, int &blk, int &seg, int &t0,
                                          int &t1) {
    if (item < w.full) {
        blk = item; seg = 0; t0 = 0; t1 = ntiles;
    } else {
        const int j = item - w.full;
        blk = w.full + j % w.rem;
        seg = j / w.rem;
        t0 = seg * w.seg_tiles;
        t1 = min(t0 + w.seg_tiles, ntiles);
    }
}

// Moment records of pass 1: [segment 0: Bpad rows][segments 1..: the rows of the segmented blocks only]; the
// returned base is indexed by the absolute spectrum row.
template <int NMOM>
__device__ __forceinline__ float *mom_segment(float *MOM, const WorkPlan &w, int seg, int Bpad, int spb = 64) {
    if (seg == 0) return MOM;
    const size_t row0 = (size_t)w.full * spb, R = (size_t)Bpad - row0;      // spb: spectra per block of the plan
    return MOM + ((size_t)Bpad + (size_t)(seg - 1) * R - row0) * NMOM;
}

// Scalar gradients (g_tau0, g_c0, g_beta: sums of cancelling per-pixel terms over every spectrum of the launch) are
// accumulated in float64 end to end: float32 inside a tile, float64 across tiles and lanes, float64 atomics across
// waves into this record (in the workspace, zeroed by k_solve), and ONE rounding to float32 per launch when the wave
// that arrives last folds the totals into the tail of the packed buffer.  (Rounds 1-2 added every wave's sum as a
// float32 atomic: up to 6 000 cancelling partials, 18x the error of a float32 numpy sum on the golden batch.)
struct Scal64 {
    double s[3];
    unsigned ticket, pad;
};
// one call per wave of the launch (any lane subset; lane 0 acts), `nwaves` calls in all
__device__ __forceinline__ void scal64_commit(Scal64 *q, double a, double b, double c, unsigned nwaves, float *accS) {
    if ((threadIdx.x & 63) != 0) return;
    if (a != 0.0) atomicAdd(&q->s[0], a);
    if (b != 0.0) atomicAdd(&q->s[1], b);
    if (c != 0.0) atomicAdd(&q->s[2], c);
    __threadfence();                                              // release: the sums before the ticket
    if (atomicAdd(&q->ticket, 1u) == nwaves - 1u) {
        __threadfence();                                          // acquire: the other waves' sums
        accS[0] += (float)__hip_atomic_load(&q->s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        accS[1] += (float)__hip_atomic_load(&q->s[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        accS[2] += (float)__hip_atomic_load(&q->s[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// scalars the kernels need, read once from device memory
struct DevConsts {
    float tau0, c0, beta;
    float t_amp, t_lscale, t_expo, t_off;   // tau(z) = t_amp * 2^(t_expo (log2(1+z) + t_lscale)) + t_off
    float offp, k1, omc0;                   // -log2(e) t_off,  -log2(e) tau0,  1 - c0   (factored-z form)
};

#ifdef QFA_PRECISE_MATH   // accuracy experiments only: libm-grade exp2/log2 and IEEE division
__device__ __forceinline__ float fast_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return log2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
__device__ __forceinline__ float fast_exp(float x) { return fast_exp2(x * QFA_LOG2E); }
__device__ __forceinline__ float fast_log(float x) { return fast_log2(x) * QFA_LN2; }

// blue-side per-element terms.  l2 = log2(1+z)
struct BlueTerms {
    float A, zd, pw, l2;
};

__device__ __forceinline__ BlueTerms blue_terms(float z, const DevConsts &k) {
    BlueTerms t;
    t.l2 = fast_log2(1.0f + z);
    float tau = k.t_amp * fast_exp2(k.t_expo * (t.l2 + k.t_lscale)) + k.t_off;   // QFA/utils.py:105-141
    t.A = fast_exp(-tau);                                                         // QFA/model.py:125
    t.pw = fast_exp2(k.beta * t.l2);                                              // (1+z)^beta, utils.py:73
    float re = 1.0f - k.c0 - fast_exp(-k.tau0 * t.pw);                            // utils.py:91
    t.zd = re * re;
    return t;
}

// ---- factored-z input form (include/qfa_hip.h, qfa_batch_t::zq1 / pix_ratio): 1 + z = zq1[s] pix_ratio[i], so
//   log2(1+z) = l2s + l2i,   (1+z)^beta = pws pwi,   A = exp(-tau(z)) = 2^(ts ti + offp)
// with per-spectrum factors ZS[s] = {ts, pws, l2s, 0} and per-pixel factors ZP[i] = {ti, pwi, l2i, 0}, both computed once
// per call in float64 (k_zfac_spec / k_zfac_pix below) and rounded once: two hardware transcendentals per blue element
// (2^x for A and for the omega term) instead of five, and none of them a logarithm (v_log_f32 is biased by -0.5 ulp,
// tools/ubench/trans_bias.hip: the zabs form carries a coherent +5e-8 bias in A that the scalar gradients see).
struct ZFac {
    float ts, pw, l2;
};
__device__ __forceinline__ ZFac zfac_load(const float4 *__restrict__ ZS, int s, bool valid) {
    ZFac z{0.f, 0.f, 0.f};
    if (valid) {
        const float4 q = ZS[s];
        z.ts = q.x; z.pw = q.y; z.l2 = q.z;
    }
    return z;
}
__device__ __forceinline__ BlueTerms blue_terms_zf(const ZFac &zs, float ti, float pwi, float l2i, const DevConsts &k) {
    BlueTerms t;
    t.l2 = zs.l2 + l2i;
    t.pw = zs.pw * pwi;                                                           // (1+z)^beta, utils.py:73
    t.A = fast_exp2(fmaf(zs.ts, ti, k.offp));                                     // QFA/model.py:125
    const float re = k.omc0 - fast_exp2(k.k1 * t.pw);                             // utils.py:91
    t.zd = re * re;
    return t;
}
// ---- resident, indexed input form (include/qfa_hip.h, ABI v3): spectrum s of the batch is row rows[s] (or s) of
// delta / error / mask, rows row_stride elements apart (the host fills in row_stride = Npix when the caller left it 0).
// The kernels address a row by a wave-uniform pointer (array base + the tile's pixel offset, SGPRs) plus a per-lane 64-bit
// element offset row * row_stride + pixel: one v_lshl_add_u64 per request, no limit on the size of the resident arrays.
__device__ __forceinline__ unsigned long long batch_row(const qfa_batch_t &bt, int s) {
    return bt.rows ? (unsigned long long)(unsigned)bt.rows[s] : (unsigned long long)(unsigned)s;
}
template <int SH>      // sbase + (off << SH): SH = 2 for the float arrays, 0 for the mask bytes
__device__ __forceinline__ const unsigned char *lane_ptr(const void *sbase, unsigned long long off) {
    return reinterpret_cast<const unsigned char *>(sbase) + (off << SH);
}

// ZS[s] from zq1[row of s] = 1 + z_qso;  ZP[i] from pix_ratio[i] = wav_i / 1215.67 (blue pixels)
__device__ __forceinline__ void zfac_spec_body(int s, const float *__restrict__ zq1, const int *__restrict__ rows,
                                               const qfa_params_t &p, const qfa_tau_t &tau, int B, float4 *__restrict__ ZS) {
    if (s >= B) return;
    const double l2s = log2((double)zq1[rows ? rows[s] : s]);
    const double ts = -1.4426950408889634 * (double)tau.amp * exp2((double)tau.expo * (l2s + log2((double)tau.scale)));
    ZS[s] = float4{(float)ts, (float)exp2((double)*p.beta * l2s), (float)l2s, 0.f};
}
static __global__ void k_zfac_spec(const float *__restrict__ zq1, const int *__restrict__ rows, qfa_params_t p, qfa_tau_t tau,
                                   int B, float4 *__restrict__ ZS) {
    zfac_spec_body(blockIdx.x * blockDim.x + threadIdx.x, zq1, rows, p, tau, B, ZS);
}
static __global__ void k_zfac_pix(const float *__restrict__ pix_ratio, qfa_params_t p, qfa_tau_t tau, int Nb,
                                  float4 *__restrict__ ZP) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Nb) return;
    const double l2i = log2((double)pix_ratio[i]);
    ZP[i] = float4{(float)exp2((double)tau.expo * l2i), (float)exp2((double)*p.beta * l2i), (float)l2i, 0.f};
}

// The per-pixel factors as the image builders take them: the table k_zfac_pix wrote, or -- the fused per-step prep kernel,
// where no table exists yet -- computed in place from pix_ratio with k_zfac_pix's own arithmetic (bit-identical images).
struct ZPSrc {
    const float4 *tab;
    const float *ratio;
    const float *beta;
    float expo;
    float offp;                 // -log2(e) tau.offset (DevConsts::offp): the image builders store it per BLUE pixel, 0 for red ones
    __device__ __forceinline__ bool on() const { return tab != nullptr || ratio != nullptr; }
    __device__ __forceinline__ float4 at(int i) const {
        if (tab) return tab[i];
        const double l2i = log2((double)ratio[i]);
        return float4{(float)exp2((double)expo * l2i), (float)exp2((double)*beta * l2i), (float)l2i, 0.f};
    }
};
__host__ __device__ inline ZPSrc zp_table(const float4 *ZP, float offp = 0.f) { return ZPSrc{ZP, nullptr, nullptr, 0.f, offp}; }

#ifndef QFA_ABL
#define QFA_ABL 0          // timing-only ablation builds (build with make -C qfa_amd/csrc B=build/var_x OUT=../libqfa_x.so EXTRA=-DQFA_ABL=n); 0 = product
#endif
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
#if QFA_ABL == 1           // no MFMA: keep operands alive, one VALU op instead
    asm volatile("" ::"v"(a), "v"(b));
    c[0] += a;
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}


// ---- float32 contractions on the bf16 XDL pipe (qfa_xdl_kernels.h explains why): x = h + m + l in bf16 pieces
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
// [15:0] = bf16(a), [31:16] = bf16(b), round to nearest even: v_cvt_pk_bf16_f32, selected by hipcc from the vector conversion.
// (Rounds 1-2 wrote the instruction as inline asm; hipcc then knows nothing about its result, which is what left the
// VALU -> MFMA hazard of section 4 of DESIGN.md uncovered and made v_dot2c_f32_bf16 behind it return stale operands.)
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2));
}
// The bf16 pairs {-1, 0} and {0, -1} in SGPRs, hidden from constant folding: hipcc encodes a known 0x0000bf80 operand of
// v_dot2c_f32_bf16 as the INLINE constant -1.0, which the instruction reads as float32 bits (0xbf800000 = {0, -1}): the wrong half.
__device__ __forceinline__ unsigned split_c_lo() {
    unsigned c;
    asm("s_mov_b32 %0, 0xbf80" : "=s"(c));
    return c;
}
__device__ __forceinline__ unsigned split_c_hi() {
    unsigned c;
    asm("s_mov_b32 %0, 0xbf800000" : "=s"(c));
    return c;
}
// x - bf16 half of a packed pair in ONE instruction: v_dot2c_f32_bf16  D = D + A.lo B.lo + A.hi B.hi  with B = {-1, 0} / {0, -1}
// (the shift / and + v_sub_f32 form costs two).  Bit-identical to the subtraction on 4.2 M random pairs with exponents
// 2^-67 .. 2^62 (round 3, on the card); below that the dot instruction flushes denormal residuals.
__device__ __forceinline__ float sub_bf16_lo(float x, unsigned packed) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, split_c_lo()), x, false);
}
__device__ __forceinline__ float sub_bf16_hi(float x, unsigned packed) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, split_c_hi()), x, false);
}
#ifndef QFA_SPLIT_DOT2
#define QFA_SPLIT_DOT2 1      // 0: residuals by shift / and + v_sub_f32 (11 instead of 7 instructions per pair of values)
#endif
// two float32 values -> three packed bf16 pairs with x = h + m + l exactly
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
#if QFA_ABL == 12          // timing only: no split arithmetic
    h = __float_as_uint(x0); m = __float_as_uint(x1); l = h ^ m;
    return;
#endif
    h = cvt_pk_bf16(x0, x1);
#if QFA_SPLIT_DOT2
    const float r0 = sub_bf16_lo(x0, h), r1 = sub_bf16_hi(x1, h);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = sub_bf16_lo(r0, m), s1 = sub_bf16_hi(r1, m);
#else
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
#endif
    l = cvt_pk_bf16(s0, s1);
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
__device__ __forceinline__ f32x4 xdl(const u32x4 &a, const u32x4 &b, f32x4 c) {          // K = 32
#if QFA_ABL == 11          // timing only: no XDL MFMA
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}
// ---- float16 pieces (round 5).  Where BOTH operands of a contraction are prepared images whose rows can be scaled by a power
// of two (the pixel image and the per-spectrum state of stage 1 of pass 2), two float16 pieces (11 + 11 bits) and THREE
// products (hh, hm, mh; mm is 2^-22) give the accuracy of the six bf16 piece products at half the MFMAs -- measured:
// tools/ubench/bf16x3_numerics.hip, profiles/r5_ubench_f16x2_numerics.txt.  float16 has 5 exponent bits: the builder scales a
// row so that its largest element is in [2^13, 2^14) (elements down to 2^-16 of it keep all 22 bits, smaller ones an absolute
// 2^-38 of it) and the consumer multiplies the power of two back in.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h(float x0, float x1, unsigned &h, unsigned &m) {      // |x| < 65 504
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 m0 = (_Float16)(x0 - (float)h0), m1 = (_Float16)(x1 - (float)h1);
    h = __builtin_bit_cast(unsigned, f16x2{h0, h1});
    m = __builtin_bit_cast(unsigned, f16x2{m0, m1});
}
__device__ __forceinline__ void split8h(const float (&x)[8], u32x4 &h, u32x4 &m) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned a, b;
        split2h(x[2 * q], x[2 * q + 1], a, b);
        h[q] = a; m[q] = b;
    }
}
__device__ __forceinline__ f32x4 xdlh(const u32x4 &a, const u32x4 &b, f32x4 c) {         // K = 32, float16 operands
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (small terms first; the first product's A operand is used again, see six_terms)
__device__ __forceinline__ f32x4 xdl3h(const u32x4 &ah, const u32x4 &am, const u32x4 &bh, const u32x4 &bm, f32x4 c) {
    c = xdlh(ah, bm, c);
    c = xdlh(am, bh, c);
    return xdlh(ah, bh, c);
}
// the power of two that brings `amax` into [2^13, 2^14) (1 for amax = 0 or not finite), and its inverse
__host__ __device__ __forceinline__ float f16_row_scale(float amax, float &inv) {
    int e = 0;
    if (amax > 0.f && amax < 3.0e38f) (void)frexpf(amax, &e); else e = 14;      // amax = m 2^e, m in [0.5, 1)
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    inv = ldexpf(1.f, e - 14);
    return ldexpf(1.f, 14 - e);
}
// The power of two for the per-element weights wD A^2 (beta of pass 2; the C / T weights of pass 1): A^2 / D <= 1 / Psi of the
// PIXEL whatever the data (D >= A^2 Psi), so s = 2^(11 + e) with Psi = m 2^e, m in [0.5, 1), keeps s wD A^2 <= 2^12 -- in any flux
// units (Psi scales with their square; tests/test_f16_dynamic_range.py).  Psi <= 0 or not finite is outside the reference's clip
// (Psi >= 1e-3, QFA/model.py:44,238) and leaves the weight without a bound: 2^-6 there (no overflow below a weight of 4e6; precision
// degrades: such a pixel's entries of the pass-1 image dominate their columns' scales).
__host__ __device__ __forceinline__ float f16_weight_scale(float psi) {
    if (!(psi > 0.f && psi < 3.0e38f)) return 0.015625f;
    int e = 0;
    (void)frexpf(psi, &e);
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    return ldexpf(1.f, 11 + e);
}
__device__ __forceinline__ f32x4 xdl16(const u32x2 &a, const u32x2 &b, f32x4 c) {        // K = 16
#if QFA_ABL == 13          // timing only: no K = 16 XDL MFMA (stage 3 of pass 2)
    asm volatile("" ::"v"(a), "v"(b));
    c[0] += 1.f;
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0,
                                                     0, 0);
}
// float32-accurate product of split operands, small terms first
template <typename V, typename F>
__device__ __forceinline__ f32x4 six_terms(F &&mm, const V &ah, const V &am, const V &al, const V &bh, const V &bm,
                                           const V &bl, f32x4 c) {
    // (the first product starts from C = 0 with a freshly allocated destination: its A operand must stay live behind
    // it -- with `al`, which is used only once, hipcc gave the destination the registers of the A operand
    // (v_mfma ... v[130:133], v[132:133], ...), and k_grads_s3 came back 2e-3 off; `ah` is used again below)
    c = mm(ah, bl, c);
    c = mm(al, bh, c);
    c = mm(am, bm, c);
    c = mm(am, bh, c);
    c = mm(ah, bm, c);
    return mm(ah, bh, c);
}
// Stage 3 of pass 2 takes the number of piece products as a template argument (TERMS) where fewer than the six of a
// float32-grade product are on offer (QFA_F_S3_FAST).  With h, m, l the bf16 pieces of an operand (|m| <= 2^-9 |h|,
// |l| <= 2^-18 |h|):  6 = all products down to 2^-18 (error ~ 2^-24);  4 = ah bh + ah bm + am bh + am bm (drops ah bl and
// al bh: <= 2 x 2^-18 per product);  3 = ah bh + ah bm + am bh (drops am bm as well: <= 3 x 2^-18 = 1.1e-5 per product).
// Six is the default everywhere (on 20 000 spectra of the headline shape the normalised F gradient is
// 2.3e-5 from the float64 oracle with six products and 5.5e-5 with three).
__device__ __forceinline__ f32x4 xdl6(const u32x4 &ah, const u32x4 &am, const u32x4 &al, const u32x4 &bh,
                                      const u32x4 &bm, const u32x4 &bl, f32x4 c) {
    return six_terms([](const u32x4 &a, const u32x4 &b, f32x4 cc) { return xdl(a, b, cc); }, ah, am, al, bh, bm, bl, c);
}
__device__ __forceinline__ f32x4 xdl16_6(const u32x2 &ah, const u32x2 &am, const u32x2 &al, const u32x2 &bh,
                                         const u32x2 &bm, const u32x2 &bl, f32x4 c) {
    return six_terms([](const u32x2 &a, const u32x2 &b, f32x4 cc) { return xdl16(a, b, cc); }, ah, am, al, bh, bm, bl, c);
}

// LDS-DMA: the wave's 64 lanes move 64 x 16 B from per-lane global addresses to lds_base + 16 * lane
__device__ __forceinline__ void glds16(const void *g, void *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

// keep a value's computation in front of this point (IR passes otherwise sink pure arithmetic past
// the sched_barrier fences of the hand-woven MFMA/VALU regions)
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }
template <typename... T>
__device__ __forceinline__ void pin(float &x, T &...rest) {
    pin(x);
    pin(rest...);
}

// Broadcast of lane l (constant after unrolling) of every 16-lane row: DPP row_newbcast, a VALU move with no LDS
// round trip (a __shfl with constant source lane compiles to ds_bpermute_b32, ~100 cycles of latency each).
__device__ __forceinline__ int dpp_row_bcast(int v, int l) {
#define QFA_BC(n) case n: return __builtin_amdgcn_mov_dpp(v, 0x150 + n, 0xf, 0xf, true);      // no `old` operand to set up
    switch (l & 15) {
        QFA_BC(0) QFA_BC(1) QFA_BC(2) QFA_BC(3) QFA_BC(4) QFA_BC(5) QFA_BC(6) QFA_BC(7)
        QFA_BC(8) QFA_BC(9) QFA_BC(10) QFA_BC(11) QFA_BC(12) QFA_BC(13) QFA_BC(14) QFA_BC(15)
    }
#undef QFA_BC
    return v;
}
// value of lane l of the KP-lane group this lane belongs to
// 32-lane groups: lane l of each half of the wave through v_readlane (scalar, no LDS round trip) + select
__device__ __forceinline__ int half_bcast(int v, int l) {
    const int a = __builtin_amdgcn_readlane(v, l & 31), b = __builtin_amdgcn_readlane(v, 32 + (l & 31));
    return (threadIdx.x & 32) ? b : a;
}
template <int KP>
__device__ __forceinline__ float group_bcast(float x, int l) {
    if constexpr (KP == 16) return __int_as_float(dpp_row_bcast(__float_as_int(x), l));
    else if constexpr (KP == 32) return __int_as_float(half_bcast(__float_as_int(x), l));
    else return __shfl(x, l, KP);
}
template <int KP>
__device__ __forceinline__ double group_bcast(double x, int l) {
    if constexpr (KP == 16) {
        const int lo = dpp_row_bcast(__double2loint(x), l), hi = dpp_row_bcast(__double2hiint(x), l);
        return __hiloint2double(hi, lo);
    } else if constexpr (KP == 32) {
        const int lo = half_bcast(__double2loint(x), l), hi = half_bcast(__double2hiint(x), l);
        return __hiloint2double(hi, lo);
    } else {
        return __shfl(x, l, KP);
    }
}

// compile-time loop: the body sees its index as a constant expression, so register arrays indexed by it are
// promoted to registers even when the loop nest is too large for the unroller to finish before SROA
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ DevConsts load_consts(const qfa_params_t &p, const qfa_tau_t &tau) {
    DevConsts k;
    k.tau0 = *p.tau0;
    k.c0 = *p.c0;
    k.beta = *p.beta;
    // log2(scale) rounded to float32 is off by up to 1.3e-7 -- the SAME error for every element of the step, i.e. a
    // coherent relative bias of up to 2.6e-7 in tau and ~1e-7 in A = exp(-tau).  Invisible in any per-element result, but
    // the scalar gradients are sums of terms that cancel 50-900x while a coherent bias adds up (tools/bias_probe.py: blue
    // sumA +4.0e-7 with the hardware log2 of the constant, round 2).  So: the float32 part of log2(scale) goes into the
    // exponent, its remainder into the amplitude, both from float64.
    const double l2s = log2((double)tau.scale);
    k.t_lscale = (float)l2s;
    k.t_amp = (float)((double)tau.amp * exp2((double)tau.expo * (l2s - (double)k.t_lscale)));
    k.t_expo = tau.expo;
    k.t_off = tau.offset;
    k.offp = -QFA_LOG2E * k.t_off;
    k.k1 = -QFA_LOG2E * k.tau0;
    k.omc0 = 1.0f - k.c0;
    return k;
}
