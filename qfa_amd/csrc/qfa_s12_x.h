// qfa_s12_x.h -- pass 2 for N_h = 17..32, stages 1 and 2 on the XDL pipe (k_s12_x).
//
// k_grads_x (qfa_grads_x.h) does not scale to KP = 32: its hand-over slots, staging buffers and a 110 KiB image ring
// need 250 KiB of LDS, and stage 3 would need Z of 16 spectra for 32 columns in registers.  Pass 2 is therefore cut
// differently there:
//   k_s12_x      stage 1 ([f^T y | f^T C^-1' f], K = 32 + 528 as 18 K-steps of six v_mfma_f32_16x16x32_bf16) and stage 2
//                (u, diag Sigma^-1, dG, the Psi / omega / tau0 / beta / c0 sums) for every (spectrum, pixel); the per-pixel
//                sums are flushed, beta = wD A^2 and gamma = A u go to HBM ([Bpad64][NpixPad] each);
//   k_grads_s3   (below) stage 3 from beta / gamma, once per 16 output columns.
// Lane layout as role A of k_grads_x: a wave = 16 spectra, lane (lo = lane & 15, g = lane >> 4) owns spectra
// s0 + 4 g + r (r = 0..3) at the pixels 32 t + 2 lo + h of half h of tile t; the A operand of stage 1 (y, C^-1' of the
// wave's spectra as three bf16 pieces, 216 registers) is loaded once per work item -- one wave per SIMD, 512 registers,
// built in the N_h > 16 translation unit.  Workgroup = 4 such waves = 64 spectra, no roles.
//
// The image of a 16-pixel half (18 K-steps x 3 pieces x 1 KiB = 54 KiB) moves through LDS in two QUARTERS of 9
// K-steps (ring of 3 x 28 KiB: a quarter + the KiB with Psi / omega of the half): a tile step is four sub-steps
// [issue the LDS-DMA of the quarter two ahead | the 54 MFMAs of this one (+ stage 2 of the half behind its second
// quarter) | counted wait for the next quarter | barrier].  The spectra go the way of k_grads_x: every wave streams the
// row segments of its 16 spectra (delta, sigma, zabs, mask) into its own LDS staging buffers by LDS-DMA TWO tiles ahead
// and copies the tile it starts on into registers.  Every request of the hot path is an asm statement, so hipcc inserts
// no vmcnt wait of its own (ordinary loads did: pass 2 at c5 5.2 ms against 4.2); the waits count the requests that
// may stay in flight (dma_wait_n), the requests per sub-step being known numbers.
#pragma once
#include "qfa_common.h"
#include "qfa_xdl_kernels.h"

#ifndef QFA_S12_ABL
#define QFA_S12_ABL 0        // timing-only ablations: 1 no beta / gamma stores, 2 no spectra loads (and their share of the waits), 4 no stage 2, 8 staging from cache
#endif
typedef float s12_f32x2 __attribute__((ext_vector_type(2)));
#ifndef QFA_S12_F16
#define QFA_S12_F16 1
#endif
template <int KP>
struct S12 {
    static constexpr int KK2 = KP * (KP + 1) / 2;
    static constexpr int NKS = 1 + (KK2 + 31) / 32;          // 18 at KP = 32
    static_assert(NKS % 2 == 0, "two quarters of NKS / 2 K-steps");
    static constexpr int NKQ = NKS / 2;                        // K-steps per quarter
    // Round 5 (QFA_S12_F16): stage 1 on TWO float16 pieces per operand and three products per K-step (qfa_common.h "float16
    // pieces"): 54 MFMAs per 16 spectra x 16 pixels instead of 108.  The image holds t f_a and t^2 f_a f_b (t: the pixel's power of
    // two, 1 / t and 1 / t^2 in the half's parameter KiB, floats 80.. and 96..); [y] and [C^-1'] (the writer: [hmean], [hcov'])
    // of a spectrum get powers of two of their own where the kernels build their A operand.
    static constexpr bool F16 = QFA_S12_F16 != 0;
    static constexpr int NP = F16 ? 2 : 3;                     // pieces per K-step
    static constexpr int KS_B = NP * 1024;
    static constexpr int PAR_IT1 = 80, PAR_IT2 = 96;           // float index in a half's parameter KiB: 1 / t, 1 / t^2 of pixel 2 lo + h
    static constexpr int Q_B = NKQ * KS_B;                     // bytes of a quarter image (18 KiB; bf16 pieces: 27)
    static constexpr int SLOT_B = Q_B + 1024;                  // ring slot: + Psi[16], omega[16] of the half (float32, one KiB)
    static constexpr int HALF_B = 2 * Q_B + 1024;              // global: [quarter 0 | Psi/omega KiB | quarter 1]
    static constexpr int TILE_B = 2 * HALF_B;                  // 110 KiB per 32-pixel tile
    static constexpr int STG_ARR = 16 * 128;                   // staging: one array of one tile, [16 rows][32 px] float
    static constexpr int STG_MASK = 3 * STG_ARR;               // mask bytes [16 rows][32 px]
    static constexpr int STG_B = 3 * STG_ARR + 512;            // delta | sigma | zabs | mask (6.5 KiB per wave and tile)
};

// ------------------------------------------------------------------------------------------------
// k_prep_s12 : F, Psi, omega -> the image above, one block per 32-pixel tile.
//   half h, K-step ks, piece, lane (g, lo), 8 k:  B[k = 32 ks + 8 g + j][px = 2 lo + h];  ks = 0: F[px][k];
//   ks >= 1: pair q = 32 (ks - 1) + 8 g + j -> F[px][a_q] F[px][b_q]
// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void k_prep_s12(const float *__restrict__ F, const float *__restrict__ Psi,
                                                  const float *__restrict__ omega, const float4 *__restrict__ ZP,
                                                  int Npix, int Nb, int Nh, unsigned char *__restrict__ IMG) {
    using X = S12<KP>;
    unsigned char *tile = IMG + (size_t)blockIdx.x * X::TILE_B;
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
        const int qt = ks / X::NKQ, kq = ks % X::NKQ;
        unsigned char *dst = tile + h * X::HALF_B + qt * (X::Q_B + 1024) + kq * X::KS_B + lane * 16;
        if constexpr (X::F16) {
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
    for (int i = threadIdx.x; i < 512; i += 256) {            // Psi, omega of each half's pixels (+ zero padding of the KiB)
        const int h = i >> 8, j = i & 255;
        float *po = reinterpret_cast<float *>(tile + h * X::HALF_B + X::Q_B);
        const int px = p0 + 2 * (j & 15) + h;
        float v = 0.f;
        if (j < 16) v = px < Npix ? Psi[px] : 0.f;
        else if (j < 32) v = px < Nb ? omega[px] : 0.f;
        else if (j < 80 && ZP && px < Nb) {                 // factored-z form: ti | pwi | l2i of the half's pixels
            const float4 q = ZP[px];
            v = j < 48 ? q.x : (j < 64 ? q.y : q.z);
        } else if (X::F16 && j >= X::PAR_IT1 && j < X::PAR_IT2 + 16) v = tsc[2 * (j & 15) + h][j < X::PAR_IT2 ? 1 : 2];
        po[j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// k_s12_x.  One work item = (block of 64 spectra, range of 32-pixel tiles).  slab != NULL: deterministic mode (the
// per-pixel sums go to row blk of the slab by plain stores, the scalar sums to slabS[item][wave][3]).
// ------------------------------------------------------------------------------------------------
template <int KP, bool HASA, bool ZF>
__global__ __launch_bounds__(256, 1) void k_s12_x(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau, int B, int Npix, int Nb,
                                                  int Nh, int ntiles, WorkPlan wp, const unsigned char *__restrict__ IMG,
                                                  const float *__restrict__ SOL, float *__restrict__ BG,
                                                  float *__restrict__ GG, int bg_stride, float *__restrict__ accum,
                                                  float *__restrict__ slab, double *__restrict__ slabS, int slab_stride,
                                                  Scal64 *__restrict__ sc64, const float4 *__restrict__ ZS) {
    using C = Cfg<KP>;
    using X = S12<KP>;
    constexpr int RING = 3;                                    // quarters in LDS: two of DMA distance (the image streams
                                                               // from the Infinity Cache: 27.5 MB at c5 against 4 MB of L2)
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING][X::SLOT_B];
    __shared__ __attribute__((aligned(16))) unsigned char lstg[4][2][X::STG_B];      // [wave][tile parity]
    __shared__ float lpsum[2][4][4][32];                       // [tile parity][wave][sumA | gPsi | gOmega | cnt][32 px]
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_uniform(tid >> 6);
    const int lo = lane & 15, g = lane >> 4;
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int s0 = blk * 64 + wv * 16;
    const bool active = s0 < B;                                // wave-uniform
    const int n = t1 - t0;
    const int nbt = (Nb + 31) >> 5;
    const DevConsts k = load_consts(p, tau);
    const bool det = slab != nullptr;
    float *accA = (det ? slab + (size_t)blk * (size_t)slab_stride : accum) + (size_t)Npix * Nh;   // sumA | gPsi | gOmega | cnt
    float *accS = accum + (size_t)Npix * Nh + 3 * (size_t)Npix + Nb;
    for (int i = tid; i < 2 * 4 * 4 * 32; i += 256) (&lpsum[0][0][0][0])[i] = 0.f;      // inactive waves' rows stay 0

    // A operand of stage 1: spectrum s0 + lo, k = 32 ks + 8 g + j
    u32x4 S1h[X::NKS], S1m[X::NKS], S1l[X::F16 ? 1 : X::NKS];
    float is0[4] = {1.f, 1.f, 1.f, 1.f}, is1[4] = {1.f, 1.f, 1.f, 1.f};      // F16: inverse powers of two of the spectra 4 g + r (y | C^-1')
    {
        const bool v = active && (s0 + lo) < B;
        const float *sol = SOL + (size_t)(v ? s0 + lo : 0) * C::NSOL;
        auto value = [&](int ks, int kk) __attribute__((always_inline)) {
            float val = 0.f;
            if (ks == 0) {
                if (v && kk < KP) val = sol[kk];
            } else {
                const int qq = 32 * (ks - 1) + kk;
                if (v && qq < X::KK2) val = sol[C::SOL_CI + qq];
            }
            return val;
        };
        float sc0 = 1.f, sc1 = 1.f;                              // powers of two of the spectrum's K-step 0 values / its pair values
        if constexpr (X::F16) {
            float m0 = 0.f, m1 = 0.f;
#pragma unroll
            for (int ks = 0; ks < X::NKS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fabsf(value(ks, 8 * g + j));
                    if (ks == 0) m0 = fmaxf(m0, a); else m1 = fmaxf(m1, a);
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
        for (int ks = 0; ks < X::NKS; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = value(ks, 8 * g + j) * (ks == 0 ? sc0 : sc1);
            if constexpr (X::F16) split8h(x, S1h[ks], S1m[ks]);
            else split8(x, S1h[ks], S1m[ks], S1l[ks]);
        }
    }
    bool sv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) sv[r] = active && (s0 + 4 * g + r) < B;
    // (a small window: the workgroups of an XCD then stream the same few tiles of the image through its L2)
    const int rot = n > 0 ? (int)(((unsigned)blk * 2654435761u) % (unsigned)min(n, 4)) : 0;
    auto tile_of = [&](int c) {
        int x = c + rot;
        if (x >= n) x -= n;
        return t0 + x;
    };

    struct Spec {                      // one lane's 4 spectra x 2 pixels of a tile; sigma < 0: masked
        float d[4][2], sg[4][2], z[4][2];
    };
    // Staging (qfa_grads_x.h): slot q of an array holds row q ^ ((q >> 2) & 1); 16-byte pieces, two instructions per
    // float array, the mask as 4-byte pieces.  Returns the requests issued: 8 (all 32 pixels inside the row and, on a blue
    // tile, inside the blue side), else 0 = "ragged: ordinary loads and LDS stores, wait for everything".
    const int last_row = active ? min(15, B - 1 - s0) : 0;          // wave-uniform
    // Spectrum s0 + r of the batch is row batch_row(s0 + r) of the batch arrays (ABI v3: rows / row_stride, qfa_common.h);
    // the rows of the lane's two 16-byte pieces per array stay in registers (one wave per SIMD: room to spare)
    const unsigned RS = (unsigned)bt.row_stride;                     // (elements; < 2^31: check_batch)
    auto slot_row = [&](int q) __attribute__((always_inline)) { return (unsigned)min(q ^ ((q >> 2) & 1), last_row); };
    unsigned R2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) R2[i] = (unsigned)batch_row(bt, (active ? s0 : 0) + (int)slot_row(8 * i + (lane >> 3)));
    auto stage_tile = [&](int tg, int par) -> int {
        if (QFA_S12_ABL & 2) return 8;
        if (QFA_S12_ABL & 8) tg = t0;                 // timing only: the staging always re-reads the item's first tile (cache hits)
        const bool zblue = !ZF && tg < nbt;                                               // wave-uniform
        const bool fast = (32 * tg + 31 < Npix) && (!zblue || 32 * tg + 31 < Nb) && !QFA_TRACKED_LOADS;
        // the third array: zabs rows (Nb apart, Nb long) on a blue tile, else the delta rows again
        const float *zb = zblue ? bt.zabs : bt.delta;
        const unsigned zpitch = zblue ? (unsigned)Nb : RS;
        const int zlen = zblue ? Nb : Npix;
        unsigned char *buf = lstg[wv][par];
        if (fast) {
            const unsigned dst = wave_uniform(lds_addr(buf));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned pc = 32u * (unsigned)tg + 4u * (unsigned)(lane & 7);       // first pixel of the piece
                const unsigned long long o = (unsigned long long)R2[i] * RS + pc;
                glds16p(lane_ptr<2>(bt.delta, o), dst + 0 * X::STG_ARR + i * 1024);
                glds16p(lane_ptr<2>(bt.error, o), dst + 1 * X::STG_ARR + i * 1024);
                if (!ZF) glds16p(lane_ptr<2>(zb, (unsigned long long)R2[i] * zpitch + pc), dst + 2 * X::STG_ARR + i * 1024);
                glds4p(lane_ptr<0>(bt.mask, o), dst + X::STG_MASK + i * 256);
            }
            return ZF ? 6 : 8;
        }
        float *sf = reinterpret_cast<float *>(buf);
        unsigned char *mb = buf + X::STG_MASK;
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            const int q = 2 * i + (lane >> 5), pxl = lane & 31;
            const unsigned long long R = batch_row(bt, (active ? s0 : 0) + (int)slot_row(q));
            const int px = 32 * tg + pxl;
            const unsigned long long o = R * RS + (unsigned)min(px, Npix - 1);
            sf[0 * (X::STG_ARR / 4) + q * 32 + pxl] = bt.delta[o];
            sf[1 * (X::STG_ARR / 4) + q * 32 + pxl] = bt.error[o];
            if (!ZF) sf[2 * (X::STG_ARR / 4) + q * 32 + pxl] = zb[R * zpitch + (unsigned)min(px, zlen - 1)];
            mb[q * 32 + pxl] = px < Npix ? bt.mask[o] : (unsigned char)0;
        }
        return 0;
    };
    auto take_tile = [&](int par, Spec &cur) {
        const unsigned char *sb = lstg[wv][par] + 8 * lo;
        const unsigned char *mb = lstg[wv][par] + X::STG_MASK + 2 * lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int slot = 4 * g + (r ^ (g & 1));
            const s12_f32x2 d2 = *reinterpret_cast<const s12_f32x2 *>(sb + 0 * X::STG_ARR + slot * 128);
            const s12_f32x2 e2 = *reinterpret_cast<const s12_f32x2 *>(sb + 1 * X::STG_ARR + slot * 128);
            s12_f32x2 z2 = {0.f, 0.f};
            if (!ZF) z2 = *reinterpret_cast<const s12_f32x2 *>(sb + 2 * X::STG_ARR + slot * 128);
            const unsigned mk = *reinterpret_cast<const unsigned short *>(mb + slot * 32);
            cur.d[r][0] = d2[0]; cur.d[r][1] = d2[1];
            cur.sg[r][0] = (mk & 0xffu) ? fabsf(e2[0]) : -1.f;        // (sign bit = masked; only sigma^2 is used)
            cur.sg[r][1] = (mk & 0xff00u) ? fabsf(e2[1]) : -1.f;
            cur.z[r][0] = z2[0]; cur.z[r][1] = z2[1];
        }
    };

    // LDS-DMA of quarter u (u = 4 c + 2 h + j: tile c, half h, K-range j) into ring slot u % RING; wave w moves the 1-KiB
    // pieces w, w + 4, ...; the quarter with j = 0 is followed by the half's Psi / omega KiB
    auto get_quarter = [&](int u) {
        const int c = u >> 2, h = (u >> 1) & 1, j = u & 1;
        const unsigned char *src = uniform_ptr(IMG + (size_t)tile_of(c) * X::TILE_B + h * X::HALF_B + j * (X::Q_B + 1024));
        unsigned char *dst = lds[u % RING];
        constexpr int NCH = X::Q_B / 1024;
#pragma unroll
        for (int i = 0; i < (NCH + 1 + 3) / 4; ++i) {
            const int ch = wv + 4 * i;
            if (ch < NCH) glds16a(src + ch * 1024, (unsigned)lane * 16u, wave_uniform(lds_addr(dst + ch * 1024)));
            else if (ch == NCH && j == 0)
                glds16a(src + X::Q_B, (unsigned)lane * 16u, wave_uniform(lds_addr(dst + X::Q_B)));
        }
    };

    double s_tau0 = 0.0, s_c0 = 0.0, s_beta = 0.0;
    f32x4 afy, aq;
    float PsiH = 0.f, omH = 0.f, tiH = 0.f, pwiH = 0.f, l2iH = 0.f, it1H = 1.f, it2H = 1.f;
    ZFac zs[4];                                   // factored-z form: per-spectrum factors of the lane's four spectra
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[r] = zfac_load(ZS, s0 + 4 * g + r, ZF && sv[r]);

    // the K-steps of quarter j of a half: B pieces of K-step ks + 1 are read while the six MFMAs of ks run
    auto quarter = [&](auto jtag, const unsigned char *img) {
        constexpr int j = decltype(jtag)::value;
        const unsigned char *bp = img + lane * 16;
        if (j == 0) {
            afy = f32x4{0.f, 0.f, 0.f, 0.f};
            aq = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *po = reinterpret_cast<const float *>(img + X::Q_B);
            PsiH = po[lo];
            omH = po[16 + lo];
            if (X::F16) { it1H = po[X::PAR_IT1 + lo]; it2H = po[X::PAR_IT2 + lo]; }
            if (ZF) { tiH = po[32 + lo]; pwiH = po[48 + lo]; l2iH = po[64 + lo]; }
        }
        constexpr int NP = X::NP;
        u32x4 bq[2][NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) bq[0][pc] = *reinterpret_cast<const u32x4 *>(bp + pc * 1024);
#pragma unroll
        for (int kq = 0; kq < X::NKQ; ++kq) {
            if (kq + 1 < X::NKQ) {
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    bq[(kq + 1) & 1][pc] = *reinterpret_cast<const u32x4 *>(bp + (kq + 1) * X::KS_B + pc * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 &bh = bq[kq & 1][0], &bm = bq[kq & 1][1], &bl = bq[kq & 1][NP - 1];
            // (ks = NKQ j + kq is a compile-time constant per (j, kq): j is passed as a literal below)
            if constexpr (X::F16) {
                if (j == 0) {
                    if (kq == 0) afy = xdl3h(S1h[0], S1m[0], bh, bm, afy);
                    else aq = xdl3h(S1h[kq], S1m[kq], bh, bm, aq);
                } else {
                    aq = xdl3h(S1h[X::NKQ + kq], S1m[X::NKQ + kq], bh, bm, aq);
                }
            } else {
                constexpr int L0 = X::F16 ? 0 : 1;
                if (j == 0) {
                    if (kq == 0) afy = xdl6(S1h[0], S1m[0], S1l[0], bh, bm, bl, afy);
                    else aq = xdl6(S1h[kq], S1m[kq], S1l[L0 * kq], bh, bm, bl, aq);
                } else {
                    aq = xdl6(S1h[X::NKQ + kq], S1m[X::NKQ + kq], S1l[L0 * (X::NKQ + kq)], bh, bm, bl, aq);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (X::F16) {
            if (j == 1) {                      // the powers of two back in: element r <-> spectrum 4 g + r, the lane's pixel
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    afy[r] = (afy[r] * is0[r]) * it1H;
                    aq[r] = (aq[r] * is1[r]) * it2H;
                }
            }
        }
    };

    // stage 2 of the lane's four elements of half h of tile tg (as role A of k_grads_x), beta / gamma to HBM, per-pixel
    // sums of the wave to its LDS row
    float t_tau0 = 0.f, t_c0 = 0.f, t_beta = 0.f;
    auto stage2 = [&](auto blue_tag, int tg, int h, const Spec &cur, int par) {
        constexpr bool BLUE = decltype(blue_tag)::value;
        const int px = 32 * tg + 2 * lo + h;
        const bool inb = px < Npix;
        const bool blue = px < Nb;
        const float Psi = PsiH, om = omH;
        float gPsi = 0.f, gOm = 0.f, sA = 0.f, cnt = 0.f;
        float betaR[4], gamR[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg = cur.sg[r][h];
            const bool wv_ = inb & sv[r] & (__float_as_int(sg) >= 0);
            const float dd = wv_ ? cur.d[r][h] : 0.f;
            if (BLUE) {
                float l2, pw, Ab, re;
                if (ZF) {                                                                         // qfa_common.h, ZFac
                    l2 = zs[r].l2 + l2iH;
                    pw = zs[r].pw * pwiH;
                    Ab = fast_exp2(fmaf(zs[r].ts, tiH, k.offp));                                  // QFA/model.py:125
                    re = k.omc0 - fast_exp2(k.k1 * pw);                                           // QFA/utils.py:91
                } else {
                    l2 = fast_log2(1.0f + cur.z[r][h]);
                    pw = fast_exp2(k.beta * l2);
                    const float tauv = k.t_amp * fast_exp2(k.t_expo * (l2 + k.t_lscale)) + k.t_off;   // QFA/utils.py:105-141
                    Ab = fast_exp2(-tauv * QFA_LOG2E);                                            // QFA/model.py:125
                    re = 1.0f - k.c0 - fast_exp2(-k.tau0 * pw * QFA_LOG2E);                       // QFA/utils.py:91
                }
                if (HASA) Ab = bt.A_blue[(size_t)(active ? min(s0 + 4 * g + r, B - 1) : 0) * Nb + min(px, Nb - 1)];
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
        }
        if (active && !(QFA_S12_ABL & 1)) {  // (bg_stride >= 32 ntiles: every pixel of the last tile has its own slot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = (size_t)(s0 + 4 * g + r) * bg_stride + px;
                BG[o] = betaR[r];
                GG[o] = gamR[r];
            }
        }
        // per-pixel sums over the wave's 16 spectra: quantity q ends up in the 16-lane row q (qfa_grads_x.h)
        {
            const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sA), __float_as_uint(gPsi), false, false);
            const auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(gOm), __float_as_uint(cnt), false, false);
            const float u01 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);
            const float u23 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);
            const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(u01), __float_as_uint(u23), false, false);
            lpsum[par][wv][g][2 * lo + h] = __uint_as_float(t[0]) + __uint_as_float(t[1]);
        }
        if (BLUE && h == 1) {
            s_tau0 += (double)t_tau0;
            s_c0 += (double)t_c0;
            s_beta += (double)t_beta;
            t_tau0 = 0.f; t_c0 = 0.f; t_beta = 0.f;
        }
    };
    // per-pixel sums [sumA | gPsi | gOmega | cnt] of tile tg: thread (which = tid >> 5, pxl = tid & 31) of the first 128.
    // Every lane issues its request (the waits count it): a lane outside the arrays adds 0 to an element inside them, a
    // different one per lane, or -- deterministic mode -- stores into the spare floats at the end of the slab row.
    float *sink = (det ? slab + (size_t)blk * (size_t)slab_stride + (slab_stride - 64) : accum) + lane;
    auto flush_P = [&](int tg, int par) {
        if ((QFA_S12_ABL & 1) || tid >= 128) return;
        const int which = tid >> 5, pxl = tid & 31;
        const float v = (lpsum[par][0][which][pxl] + lpsum[par][1][which][pxl]) +
                        (lpsum[par][2][which][pxl] + lpsum[par][3][which][pxl]);
        const int px = 32 * tg + pxl;
        const bool ok = (px < Npix) & ((which != 2) | (px < Nb));
        const int pxc = min(px, Npix - 1);
        // (a red pixel's lane of the gOmega group adds 0 to the pixel's count instead)
        const int offc = (which == 2 && pxc >= Nb) ? 2 * Npix + Nb + pxc : which * Npix - (which == 3 ? Npix - Nb : 0) + pxc;
        if (det) *(ok ? accA + offc : sink) = v;
        else atomicAdd(accA + offc, ok ? v : 0.f);
    };

    if (n > 0) {
        // requests a sub-step puts into this wave's queue BEHIND its image DMA: the flush atomic of (h0, j0) (waves 0, 1),
        // the staging of tile c + 2 issued there, the eight beta / gamma stores of a sub-step with stage 2
        int stag_a = 0, stag_b = 0;                 // staging requests in flight for buffer 0 / 1 (0: not a counted issue)
        bool drain = false;                         // a ragged staging is pending: wait for everything
        if (active) {
            stag_a = stage_tile(tile_of(0), 0);
            if (n > 1) stag_b = stage_tile(tile_of(1), 1);
        }
        get_quarter(0);
        if (4 * n > 1) get_quarter(1);
        dma_wait<0>();
        step_barrier();
        // pieces of quarter u moved by this wave (get_quarter: wave w moves the pieces w, w + 4, ...; piece NCH = the parameter KiB, j = 0)
        constexpr int NCHQ = X::Q_B / 1024;
        auto cnt_q = [&](int u) { return (NCHQ - wv + 3) / 4 + ((NCHQ % 4 == wv && !(u & 1)) ? 1 : 0); };
        int rest_prev = 0;                          // requests of the previous sub-step behind its image DMA
        Spec cur;
        auto tile_step = [&](int c, int &stag_cur) {
            const int tg = tile_of(c);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int u = 4 * c + 2 * h + j;
                    const bool more = u + 2 < 4 * n;
                    if (more) get_quarter(u + 2);                 // (slot of quarter u - 1: free behind the barrier)
                    int rest = 0;
                    if (h == 0 && j == 0) {
                        if (c >= 1) {
                            flush_P(tile_of(c - 1), (c - 1) & 1);
                            rest += (!(QFA_S12_ABL & 1) && tid < 128) ? 1 : 0;
                        }
                        if (active) {
                            take_tile(c & 1, cur);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // staging buffer read: it may be overwritten
                            stag_cur = 0;
                            if (c + 2 < n) {
                                stag_cur = stage_tile(tile_of(c + 2), c & 1);
                                if (stag_cur == 0) drain = true;
                                rest += stag_cur;
                            }
                        }
                    }
                    if (active) {
                        if (j == 0) quarter(std::integral_constant<int, 0>{}, lds[u % RING]);
                        else quarter(std::integral_constant<int, 1>{}, lds[u % RING]);
                        if (j == 1 && !(QFA_S12_ABL & 4)) {
                            if (tg < nbt) stage2(std::true_type{}, tg, h, cur, c & 1);
                            else stage2(std::false_type{}, tg, h, cur, c & 1);
                            rest += (QFA_S12_ABL & 1) ? 0 : 8;
                        }
                    }
                    // the DMA of quarter u + 1 (issued at the start of the previous sub-step) must have landed; behind it in
                    // the queue: the rest of the previous sub-step, this sub-step's DMA and its rest
                    if (!more || drain) dma_wait<0>();
                    else dma_wait_n(rest_prev + cnt_q(u + 2) + rest);
                    if (drain && (h == 1 && j == 1)) drain = false;       // (everything has been waited for since)
                    rest_prev = rest;
                    step_barrier();
                }
            }
        };
        for (int c = 0; c < n; c += 2) {
            tile_step(c, stag_a);
            if (c + 1 < n) tile_step(c + 1, stag_b);
        }
        flush_P(tile_of(n - 1), (n - 1) & 1);
    }
    if (active) {
        for (int o = 32; o >= 1; o >>= 1) {
            s_tau0 += __shfl_xor(s_tau0, o);
            s_c0 += __shfl_xor(s_c0, o);
            s_beta += __shfl_xor(s_beta, o);
        }
    }
    if (lane == 0) {
        if (det) {
            double *q = slabS + ((size_t)blockIdx.x * 4 + wv) * 3;
            q[0] = active ? s_tau0 : 0.0; q[1] = active ? s_c0 : 0.0; q[2] = active ? s_beta : 0.0;
        } else {
            scal64_commit(sc64, active ? s_tau0 : 0.0, active ? s_c0 : 0.0, active ? s_beta : 0.0, gridDim.x * 4u, accS);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_predict_x32 (N_h = 17..32): the posterior writer on the XDL pipe -- cont = F hmean + mu, unc = sqrt(f^T hcov f) for
// ALL pixels of every spectrum (reference QFA/model.py:177-180).  It is stage 1 of k_s12_x with [hmean | hcov'] (as
// k_solve<KP, true> leaves them in SOL) in the place of [y | C^-1']: same image (k_prep_s12 called with mu in the place
// of Psi, so the half's KiB carries mu of the lane's pixel), same quarter ring, same counted waits; no spectra are read.
// Lane (lo, g) owns the pixels 32 t + 2 lo + h of the spectra s0 + 4 g + r: one 8-byte store per spectrum row and
// array behind the second half (128 contiguous bytes per row and wave instruction), eight requests per tile.
// ------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) s12_pair { float v[2]; };       // 4-byte aligned 8-byte store
typedef float s12_v2 __attribute__((ext_vector_type(2), aligned(4)));
// (non-temporal, like k_predict_x's: the prediction outputs are not read back by the call)
__device__ __forceinline__ void s12_store2(float *dst, float a, float b) { __builtin_nontemporal_store(s12_v2{a, b}, reinterpret_cast<s12_v2 *>(dst)); }
template <int KP>
__global__ __launch_bounds__(256, 1) void k_predict_x32(int B, int Npix, int ntiles, WorkPlan wp,
                                                        const unsigned char *__restrict__ IMG,
                                                        const float *__restrict__ SOL, float *__restrict__ cont,
                                                        float *__restrict__ unc) {
    using C = Cfg<KP>;
    using X = S12<KP>;
    constexpr int RING = 3;
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING][X::SLOT_B];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_uniform(tid >> 6);
    const int lo = lane & 15, g = lane >> 4;
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int s0 = blk * 64 + wv * 16;
    const bool active = s0 < B;                                // wave-uniform
    const int n = t1 - t0;
    if (n <= 0) return;

    u32x4 S1h[X::NKS], S1m[X::NKS], S1l[X::F16 ? 1 : X::NKS];   // A operand: spectrum s0 + lo, k = 32 ks + 8 g + j
    float is0[4] = {1.f, 1.f, 1.f, 1.f}, is1[4] = {1.f, 1.f, 1.f, 1.f};      // F16: inverse powers of two of the spectra 4 g + r (hmean | hcov')
    {
        const bool v = active && (s0 + lo) < B;
        const float *sol = SOL + (size_t)(v ? s0 + lo : 0) * C::NSOL;
        auto value = [&](int ks, int kk) __attribute__((always_inline)) {
            float val = 0.f;
            if (ks == 0) {
                if (v && kk < KP) val = sol[kk];
            } else {
                const int qq = 32 * (ks - 1) + kk;
                if (v && qq < X::KK2) val = sol[C::SOL_CI + qq];
            }
            return val;
        };
        float sc0 = 1.f, sc1 = 1.f;                              // powers of two of the spectrum's K-step 0 values / its pair values
        if constexpr (X::F16) {
            float m0 = 0.f, m1 = 0.f;
#pragma unroll
            for (int ks = 0; ks < X::NKS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fabsf(value(ks, 8 * g + j));
                    if (ks == 0) m0 = fmaxf(m0, a); else m1 = fmaxf(m1, a);
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
        for (int ks = 0; ks < X::NKS; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = value(ks, 8 * g + j) * (ks == 0 ? sc0 : sc1);
            if constexpr (X::F16) split8h(x, S1h[ks], S1m[ks]);
            else split8(x, S1h[ks], S1m[ks], S1l[ks]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the operand loads above are the only tracked loads)
    const bool full_wave = active && s0 + 16 <= B;
    const int rot = (int)(((unsigned)blk * 2654435761u) % (unsigned)min(n, 4));
    auto tile_of = [&](int c) {
        int x = c + rot;
        if (x >= n) x -= n;
        return t0 + x;
    };
    auto get_quarter = [&](int u) {
        const int c = u >> 2, h = (u >> 1) & 1, j = u & 1;
        const unsigned char *src = uniform_ptr(IMG + (size_t)tile_of(c) * X::TILE_B + h * X::HALF_B + j * (X::Q_B + 1024));
        unsigned char *dst = lds[u % RING];
        constexpr int NCH = X::Q_B / 1024;
#pragma unroll
        for (int i = 0; i < (NCH + 1 + 3) / 4; ++i) {
            const int ch = wv + 4 * i;
            if (ch < NCH) glds16a(src + ch * 1024, (unsigned)lane * 16u, wave_uniform(lds_addr(dst + ch * 1024)));
            else if (ch == NCH && j == 0)
                glds16a(src + X::Q_B, (unsigned)lane * 16u, wave_uniform(lds_addr(dst + X::Q_B)));
        }
    };
    f32x4 afy, aq;
    float muH = 0.f, it1H = 1.f, it2H = 1.f;
    auto quarter = [&](auto jtag, const unsigned char *img) {
        constexpr int j = decltype(jtag)::value;
        const unsigned char *bp = img + lane * 16;
        if (j == 0) {
            afy = f32x4{0.f, 0.f, 0.f, 0.f};
            aq = f32x4{0.f, 0.f, 0.f, 0.f};
            muH = reinterpret_cast<const float *>(img + X::Q_B)[lo];
            if (X::F16) { it1H = reinterpret_cast<const float *>(img + X::Q_B)[X::PAR_IT1 + lo]; it2H = reinterpret_cast<const float *>(img + X::Q_B)[X::PAR_IT2 + lo]; }
        }
        constexpr int NP = X::NP;
        u32x4 bq[2][NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) bq[0][pc] = *reinterpret_cast<const u32x4 *>(bp + pc * 1024);
#pragma unroll
        for (int kq = 0; kq < X::NKQ; ++kq) {
            if (kq + 1 < X::NKQ) {
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    bq[(kq + 1) & 1][pc] = *reinterpret_cast<const u32x4 *>(bp + (kq + 1) * X::KS_B + pc * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 &bh = bq[kq & 1][0], &bm = bq[kq & 1][1], &bl = bq[kq & 1][NP - 1];
            if constexpr (X::F16) {
                if (j == 0) {
                    if (kq == 0) afy = xdl3h(S1h[0], S1m[0], bh, bm, afy);
                    else aq = xdl3h(S1h[kq], S1m[kq], bh, bm, aq);
                } else {
                    aq = xdl3h(S1h[X::NKQ + kq], S1m[X::NKQ + kq], bh, bm, aq);
                }
            } else {
                constexpr int L0 = X::F16 ? 0 : 1;
                if (j == 0) {
                    if (kq == 0) afy = xdl6(S1h[0], S1m[0], S1l[0], bh, bm, bl, afy);
                    else aq = xdl6(S1h[kq], S1m[kq], S1l[L0 * kq], bh, bm, bl, aq);
                } else {
                    aq = xdl6(S1h[X::NKQ + kq], S1m[X::NKQ + kq], S1l[L0 * (X::NKQ + kq)], bh, bm, bl, aq);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (X::F16) {
            if (j == 1) {                      // the powers of two back in: element r <-> spectrum 4 g + r, the lane's pixel
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    afy[r] = (afy[r] * is0[r]) * it1H;
                    aq[r] = (aq[r] * is1[r]) * it2H;
                }
            }
        }
    };

    get_quarter(0);
    if (4 * n > 1) get_quarter(1);
    dma_wait<0>();
    step_barrier();
    constexpr int NCHQ = X::Q_B / 1024;            // pieces of quarter u moved by this wave (see k_s12_x)
    auto cnt_q = [&](int u) { return (NCHQ - wv + 3) / 4 + ((NCHQ % 4 == wv && !(u & 1)) ? 1 : 0); };
    int rest_prev = 0;                              // stores of the previous sub-step, behind its image DMA in the queue
    float co0[4], un0[4];
    for (int c = 0; c < n; ++c) {
        const int tg = tile_of(c);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int u = 4 * c + 2 * h + j;
                const bool more = u + 2 < 4 * n;
                if (more) get_quarter(u + 2);
                int rest = 0;
                bool drain = false;
                if (active) {
                    if (j == 0) quarter(std::integral_constant<int, 0>{}, lds[u % RING]);
                    else quarter(std::integral_constant<int, 1>{}, lds[u % RING]);
                    if (j == 1 && h == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            co0[r] = afy[r] + muH;
                            un0[r] = __builtin_amdgcn_sqrtf(aq[r]);
                        }
                    }
                    if (j == 1 && h == 1) {
                        const int px = 32 * tg + 2 * lo;
                        if (full_wave && 32 * tg + 31 < Npix) {          // wave-uniform: exactly eight store instructions
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const size_t o = (size_t)(s0 + 4 * g + r) * Npix + px;
                                s12_store2(cont + o, co0[r], afy[r] + muH);
                                s12_store2(unc + o, un0[r], __builtin_amdgcn_sqrtf(aq[r]));
                            }
                            rest = 8;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int sp = s0 + 4 * g + r;
                                if (sp < B && px < Npix) {
                                    cont[(size_t)sp * Npix + px] = co0[r];
                                    unc[(size_t)sp * Npix + px] = un0[r];
                                }
                                if (sp < B && px + 1 < Npix) {
                                    cont[(size_t)sp * Npix + px + 1] = afy[r] + muH;
                                    unc[(size_t)sp * Npix + px + 1] = __builtin_amdgcn_sqrtf(aq[r]);
                                }
                            }
                            drain = true;
                        }
                    }
                }
                // the DMA of quarter u + 1 (issued at the start of the previous sub-step) must have landed; behind it in the
                // queue: the stores of the previous sub-step, this sub-step's DMA and its stores
                if (!more || drain) dma_wait<0>();
                else dma_wait_n(rest_prev + cnt_q(u + 2) + rest);
                rest_prev = drain ? 0 : rest;
                step_barrier();
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_grads_s3 (N_h = 17..32): stage 3 of pass 2 alone, for the output columns 16 bhalf .. 16 bhalf + 15, from the
// beta / gamma k_s12_x (or k_grads) stored:  accF[px][b] += sum_s beta_{s,px} (F_tile Z_s)[px][b] + sum_s gamma_{s,px} p_s[b].
// Work items and flush as k_grads (16-pixel tiles, 4 waves = 64 spectra, per-wave LDS slots summed in fixed order);
// lane (px = lane & 15, g = lane >> 4): rows 4 g + r of the products.  Z_s (K = a = 32) as two bf16 pieces of all 16
// spectra in 128 registers, TERMS piece products per spectrum (template argument: four by default, three with QFA_F_S3_FAST); 54 MFMAs and 64 FMAs per tile, no transcendental.
// Inputs of a tile arrive by LDS-DMA TWO tiles ahead into the wave's own buffers (ring of 3 x 4 KiB): the beta and gamma
// tiles of its 16 spectra ([s][16 px] float, 64-byte row segments: one instruction each, and already the layout the
// beta-scaling reads) and the two F pieces (1 KiB each).  A wave's queue holds those four requests per tile and its one
// flush request; the wait at the start of a tile leaves the ten youngest in flight.
// ------------------------------------------------------------------------------------------------
#ifndef QFA_S3_SETPRIO
#define QFA_S3_SETPRIO 2     // s_setprio around the products of a tile in k_grads_s3 (c5 pass 2 3.11 / 3.18 -> 3.08 / 3.14 ms, same box)
#endif
#ifndef QFA_S3_ABL
#define QFA_S3_ABL 0         // timing-only ablations of k_grads_s3: 1 no flush, 2 no input DMA, 4 no beta-scaled products (gamma term only)
#endif
template <int KP, int TERMS>      // TERMS: bf16 piece products per stage-3 contraction -- 6 (default: float32 grade; the third
                                  // piece of Z is 64 more registers: one workgroup per CU), 4 or 3 (QFA_F_S3_FAST)
__global__ __launch_bounds__(256, 2) void k_grads_s3(int B, int Npix, int Nh, int ntiles, WorkPlan wp, int bhalf,
                                                     const float *__restrict__ PFT, const float *__restrict__ SOL,
                                                     const float *__restrict__ BG, const float *__restrict__ GG,
                                                     int bg_stride, float *__restrict__ accum, float *__restrict__ slab,
                                                     int slab_stride) {
    static_assert(KP == 32, "k_grads_s3: N_h = 17..32");
    using C = Cfg<KP>;
    // Round 5 (QFA_S3_F16, TERMS == 6 only): G_s = F_tile Z_s on TWO float16 pieces per operand, three products.  Both operands are
    // prepared: F t_px (k_prep_pf; 1 / t_px rides as a fifth KiB) and Z_s 2^k of the wave's own spectra (scaled by their largest
    // element below); beta multiplies G in float32 behind the MFMA as before, so nothing here depends on the range of the data.
    constexpr bool F16 = QFA_S3_F16 != 0 && TERMS == 6;
    constexpr int RING = 3, BUF_B = TERMS == 6 ? 5120 : 4096;   // per wave and tile: beta 1 KiB | gamma 1 KiB | Fh 1 KiB | Fm 1 KiB (| Fl 1 KiB; F16: 1 / t)
    constexpr int NREQ = TERMS == 6 ? 5 : 4;                    // input requests per wave and tile
    __shared__ __attribute__((aligned(16))) unsigned char lin[4][RING][BUF_B];
    __shared__ float ldspart[2][4][256];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_uniform(tid >> 6);
    const int lo = lane & 15, g = lane >> 4;
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int s0 = (blk * 4 + wv) * 16;
    const bool active = s0 < B;
    const int n = t1 - t0;
    const bool det = slab != nullptr;
    float *accF = det ? slab + (size_t)blk * (size_t)slab_stride : accum;
    float *sink = (det ? accF + (slab_stride - 64) : accum) + lane;
    for (int i = tid; i < 2 * 4 * 256; i += 256) (&ldspart[0][0][0])[i] = 0.f;      // inactive waves' slots stay 0

    // B operands: Z_s[a = 8g + j][col] of all 16 spectra as two bf16 pieces, p of the spectra 4g + j (gamma term)
    u32x4 Zh[16], Zm[16], Zl[(TERMS == 6 && !F16) ? 16 : 1];
    float iz[16];                                               // F16: 1 / scale of Z_s (wave-uniform)
    u32x2 ph, pm, pl;
    {
        const int zcol = 16 * bhalf + lo;
        float zraw[16][8];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bool v = active && (s0 + s) < B && zcol < KP;
            const float *sol = SOL + (size_t)(v ? s0 + s : 0) * C::NSOL + C::SOL_Z + zcol;
#pragma unroll
            for (int j = 0; j < 8; ++j) zraw[s][j] = v ? sol[(8 * g + j) * KP] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if constexpr (F16) {
                float mx = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(zraw[s][j]));
#pragma unroll
                for (int o = 1; o <= 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));      // the 32 x 16 block of Z_s this wave holds
                const float sc = f16_row_scale(mx, iz[s]);
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = zraw[s][j] * sc;
                split8h(x, Zh[s], Zm[s]);
            } else {
                iz[s] = 1.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned h, m, l;
                    split2(zraw[s][2 * q], zraw[s][2 * q + 1], h, m, l);
                    Zh[s][q] = h;
                    Zm[s][q] = m;
                    if constexpr (TERMS == 6) Zl[F16 ? 0 : s][q] = l;
                }
            }
        }
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool v = active && (s0 + 4 * g + r) < B && zcol < KP;
            pr[r] = v ? SOL[(size_t)(s0 + 4 * g + r) * C::NSOL + C::SOL_P + zcol] : 0.f;
        }
        unsigned h0, m0, l0, h1, m1, l1;
        split2(pr[0], pr[1], h0, m0, l0);
        split2(pr[2], pr[3], h1, m1, l1);
        ph = u32x2{h0, h1}; pm = u32x2{m0, m1}; pl = u32x2{l0, l1};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the operand loads above are the only tracked ones)
    const int rot = n > 0 ? (int)(((unsigned)blk * 2654435761u) % (unsigned)min(n, 32)) : 0;
    auto tile_of = [&](int c) {
        int x = c + rot;
        if (x >= n) x -= n;
        return t0 + x;
    };
    // the four input requests of tile tg: lane = (row = lane >> 2, 16-byte piece = lane & 3) of the 64-byte row segments
    const float *bgw = uniform_ptr(BG + (size_t)(active ? s0 : 0) * bg_stride);
    const float *ggw = uniform_ptr(GG + (size_t)(active ? s0 : 0) * bg_stride);
    auto get_tile = [&](int c) {
        if ((QFA_S3_ABL & 2) && c > 0) return;
        const int tg = tile_of(c);
        const unsigned dst = wave_uniform(lds_addr(lin[wv][c % RING]));
        const unsigned o = 4u * ((unsigned)(lane >> 2) * (unsigned)bg_stride + 16u * (unsigned)tg + 4u * (unsigned)(lane & 3));
        glds16a(bgw, o, dst);
        glds16a(ggw, o, dst + 1024);
        const float *fg = uniform_ptr(PFT + (size_t)tg * C::TILE_PFT + C::PFT_MAIN);
        if constexpr (F16) {                    // float16 pieces h, m and the KiB with 1 / t_px
            glds16a(fg + C::PFT_F16H, (unsigned)lane * 16u, dst + 2048);
            glds16a(fg + C::PFT_F16M, (unsigned)lane * 16u, dst + 3072);
            glds16a(fg + C::PFT_F16IT, (unsigned)lane * 16u, dst + 4096);
        } else {
            glds16a(fg, (unsigned)lane * 16u, dst + 2048);
            glds16a(fg + 256, (unsigned)lane * 16u, dst + 3072);
            if constexpr (TERMS == 6) glds16a(fg + 512, (unsigned)lane * 16u, dst + 4096);
        }
    };
    auto flush = [&](int tg, const float (*pp)[256]) {
        const int idx = lane + 64 * wv;
        const float v = (pp[0][idx] + pp[1][idx]) + (pp[2][idx] + pp[3][idx]);
        const int px = 16 * tg + (idx >> 4), b = 16 * bhalf + (idx & 15);
        const bool ok = (b < Nh) & (px < Npix);
        // every lane issues its request (the wait counts it): a lane outside the array adds 0 inside it / stores to the sink
        float *q = accF + (size_t)min(px, Npix - 1) * Nh + (b < Nh ? b : b % Nh);
        if (det) *(ok ? q : sink) = v;
        else atomicAdd(q, ok ? v : 0.f);
    };
    if (n <= 0) return;
    get_tile(0);
    if (n > 1) get_tile(1);
    if (n > 1) dma_wait<NREQ>();                                // tile 0 has landed (tile 1 may be in flight)
    else dma_wait<0>();
    __syncthreads();                                            // (also the zeroing of ldspart)
    for (int c = 0; c < n; ++c) {
        const int pbuf = c & 1;
        if (c + 2 < n) get_tile(c + 2);
        if (active) {
            const unsigned char *in = lin[wv][c % RING];
            const float *bet = reinterpret_cast<const float *>(in);           // [s][16 px]
            const float *gam = reinterpret_cast<const float *>(in + 1024);

            const u32x4 Fh = *reinterpret_cast<const u32x4 *>(in + 2048 + lane * 16),
                        Fm = *reinterpret_cast<const u32x4 *>(in + 3072 + lane * 16);
            u32x4 Fl = Fh;
            if constexpr (TERMS == 6 && !F16) Fl = *reinterpret_cast<const u32x4 *>(in + 4096 + lane * 16);
            float itp[4] = {1.f, 1.f, 1.f, 1.f};                              // F16: 1 / t of the pixels 4 g + r of the tile
            if constexpr (F16) {
                const float4 q4 = *reinterpret_cast<const float4 *>(in + 4096 + 16 * g);
                itp[0] = q4.x; itp[1] = q4.y; itp[2] = q4.z; itp[3] = q4.w;
            }

            float *part = ldspart[pbuf][wv];
            if constexpr (QFA_S3_SETPRIO != 0) __builtin_amdgcn_s_setprio(QFA_S3_SETPRIO);
            unsigned h0, m0, l0, h1, m1, l1;
            split2(gam[(4 * g + 0) * 16 + lo], gam[(4 * g + 1) * 16 + lo], h0, m0, l0);
            split2(gam[(4 * g + 2) * 16 + lo], gam[(4 * g + 3) * 16 + lo], h1, m1, l1);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 gterm = xdl16_6(u32x2{h0, h1}, u32x2{m0, m1}, u32x2{l0, l1}, ph, pm, pl, zero);
            f32x4 acc = F16 ? zero : gterm;
            const float4 *brow = reinterpret_cast<const float4 *>(bet) + g;          // beta[s][px = 4g .. 4g + 3]
#pragma unroll
            for (int s = 0; s < ((QFA_S3_ABL & 4) ? 0 : 16); ++s) {
                f32x4 G;
                if constexpr (F16) {
                    G = xdl3h(Fh, Fm, Zh[s], Zm[s], zero);
                    const float4 bq = brow[s * 4];
                    const float z = iz[s];
                    acc[0] = fmaf(bq.x * z, G[0], acc[0]);
                    acc[1] = fmaf(bq.y * z, G[1], acc[1]);
                    acc[2] = fmaf(bq.z * z, G[2], acc[2]);
                    acc[3] = fmaf(bq.w * z, G[3], acc[3]);
                    continue;
                } else if constexpr (TERMS == 6) G = xdl6(Fh, Fm, Fl, Zh[s], Zm[s], Zl[F16 ? 0 : s], zero);
                else {
                    G = xdl(Fm, Zh[s], zero);
                    if (TERMS >= 4) G = xdl(Fm, Zm[s], G);
                    G = xdl(Fh, Zm[s], G);
                    G = xdl(Fh, Zh[s], G);
                }
                const float4 bq = brow[s * 4];
                acc[0] = fmaf(bq.x, G[0], acc[0]);
                acc[1] = fmaf(bq.y, G[1], acc[1]);
                acc[2] = fmaf(bq.z, G[2], acc[2]);
                acc[3] = fmaf(bq.w, G[3], acc[3]);
            }
            if constexpr (F16) {                   // 1 / t of the row's pixel, then the (unscaled) gamma term
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) acc[rr] = fmaf(acc[rr], itp[rr], gterm[rr]);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) part[(4 * g + rr) * 16 + lo] = acc[rr];
            if constexpr (QFA_S3_SETPRIO != 0) __builtin_amdgcn_s_setprio(0);
        }
        // The wave's inputs of tile c + 1 must have landed before it goes round; behind them in its queue are the
        // flush of tile c - 1 and the requests of tile c + 2, which stay in flight.
        if (QFA_S3_ABL & 3) dma_wait<0>();
        else if (c + 1 < n) dma_wait_n((c + 2 < n ? NREQ : 0) + (c >= 1 ? 1 : 0));
        step_barrier();
        if (!(QFA_S3_ABL & 1)) flush(tile_of(c), ldspart[pbuf]);          // ldspart[pbuf] is rewritten two tiles later, behind the next barrier
    }
}
