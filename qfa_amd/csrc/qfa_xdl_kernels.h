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

template <int KP>
struct XCfg {
    using C = Cfg<KP>;
    static constexpr int NCOL = C::FW + C::PW;               // f columns first, then the pair columns
    static constexpr int PSTR = NCOL * 64;                    // bytes of one piece: 32 px x bf16 per column
    static constexpr int OFF_PSI = 3 * PSTR;                  // float32 Psi[32], omega[32]
    static constexpr int TILE_B = (3 * PSTR + 256 + 1023) / 1024 * 1024;   // whole 1-KiB LDS-DMA pieces
    static constexpr int NCHUNK = TILE_B / 1024;
};

// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void k_prep_pfx(const float *__restrict__ F, const float *__restrict__ Psi,
                                                  const float *__restrict__ omega, int Npix, int Nb, int Nh,
                                                  unsigned char *__restrict__ PFX) {
    using C = Cfg<KP>;
    using X = XCfg<KP>;
    unsigned char *tile = PFX + (size_t)blockIdx.x * X::TILE_B;
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
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 32 * blockIdx.x + q + e;
            if (okc && i < Npix) v[e] = pairc ? F[(size_t)i * Nh + a] * F[(size_t)i * Nh + b] : F[(size_t)i * Nh + a];
        }
        unsigned h, m, l;
        split2(v[0], v[1], h, m, l);
        // lane-linear for the B-operand read: [16-column tile][8-pixel group][column][8 px]
        unsigned *dst = reinterpret_cast<unsigned *>(tile + (c >> 4) * 1024 + (q >> 3) * 256 + (c & 15) * 16 + (q & 7) * 2);
        dst[0] = h;
        dst[X::PSTR / 4] = m;
        dst[2 * X::PSTR / 4] = l;
    }
    float *po = reinterpret_cast<float *>(tile + X::OFF_PSI);
    for (int idx = threadIdx.x; idx < (X::TILE_B - X::OFF_PSI) / 4; idx += 256) {
        const int i = 32 * blockIdx.x + (idx & 31);
        float v = 0.f;
        if (idx < 32) v = i < Npix ? Psi[i] : 0.f;
        else if (idx < 64) v = i < Nb ? omega[i] : 0.f;
        po[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// k_moments_x (pass 1).  Lane (sl = lane&15, g = lane>>4) owns spectrum s0+sl at the pixels 32*tile + 8g + j
// (j = 0..7): the A-operand layout A[row = lane&15][k = 8(lane>>4) + j] of v_mfma_f32_16x16x32_bf16, and 32
// contiguous bytes per input array.  B[k][col] = image column `col` at the same 8 pixels: one ds_read_b128 per
// piece (lanes of a wave read 1 KiB contiguously).  The 31-KiB image tile of the next step is moved by LDS-DMA
// (no staging registers) while the current one is consumed; one barrier per tile.
// Red tiles first (A = 1: T and b2 receive what C and b receive), then the blue tiles.
// ------------------------------------------------------------------------------------------------
struct SpecRegsX {
    float d[8], sg[8], z[8];
    unsigned m0, m1;      // 8 mask bytes
};

template <int KP, bool PREDICT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void k_moments_x(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau,
                                                      const float *__restrict__ mu, int B, int Bpad, int Npix, int Nb,
                                                      int ntiles, WorkPlan wp, const unsigned char *__restrict__ PFX,
                                                      float *__restrict__ MOM) {
    using C = Cfg<KP>;
    using X = XCfg<KP>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][X::TILE_B];
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
    const float *dbase = bt.delta + (size_t)(active ? s0 : 0) * Npix;
    const float *ebase = bt.error + (size_t)(active ? s0 : 0) * Npix;
    const uint8_t *mbase = bt.mask + (size_t)(active ? s0 : 0) * Npix;
    const float *zbase = bt.zabs + (size_t)(active ? s0 : 0) * Nb;
    const float *abase = bt.A_blue ? bt.A_blue + (size_t)(active ? s0 : 0) * Nb : nullptr;
    const int offN = srow * Npix, offB = srow * Nb;

    f32x4 accC[C::NT], accT[C::NT], accb[C::NFT], accb2[C::NFT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) accC[t] = accT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < C::NFT; ++t) accb[t] = accb2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    double qd = 0.0, ld = 0.0;        // float32 inside an 8-pixel group, float64 across groups
    float cn = 0.f, cblue = 0.f;

    // LDS-DMA of one image tile: wave w moves the 1-KiB pieces w, w+NW, ...
    auto stage = [&](int tg, int buf) {
        const unsigned char *src = PFX + (size_t)tg * X::TILE_B + lane * 16;
#pragma unroll
        for (int i = 0; i < (X::NCHUNK + NW - 1) / NW; ++i) {
            const int ch = wv + NW * i;
            if (ch < X::NCHUNK) glds16(src + ch * 1024, &lds[buf][ch * 1024]);
        }
    };

    auto run = [&](auto blue_tag, int ta, int tb) {
        constexpr bool BLUE = decltype(blue_tag)::value;
        const int n = tb - ta;
        if (n <= 0) return;                                       // block-uniform

        auto load_spec = [&](int tg, SpecRegsX &r) {
            const int pb = 32 * tg + 8 * g;
            if (pb + 7 < Npix) {
                const f4u vd0 = *reinterpret_cast<const f4u *>(dbase + offN + pb);
                const f4u vd1 = *reinterpret_cast<const f4u *>(dbase + offN + pb + 4);
                const f4u ve0 = *reinterpret_cast<const f4u *>(ebase + offN + pb);
                const f4u ve1 = *reinterpret_cast<const f4u *>(ebase + offN + pb + 4);
                const u4u vm0 = *reinterpret_cast<const u4u *>(mbase + offN + pb);
                const u4u vm1 = *reinterpret_cast<const u4u *>(mbase + offN + pb + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r.d[e] = vd0.v[e]; r.d[4 + e] = vd1.v[e];
                    r.sg[e] = ve0.v[e]; r.sg[4 + e] = ve1.v[e];
                }
                r.m0 = (unsigned)vm0.v[0] | ((unsigned)vm0.v[1] << 8) | ((unsigned)vm0.v[2] << 16) |
                       ((unsigned)vm0.v[3] << 24);
                r.m1 = (unsigned)vm1.v[0] | ((unsigned)vm1.v[1] << 8) | ((unsigned)vm1.v[2] << 16) |
                       ((unsigned)vm1.v[3] << 24);
            } else {                                              // ragged end of the pixel axis
                r.m0 = r.m1 = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int px = min(pb + e, Npix - 1);
                    r.d[e] = dbase[offN + px];
                    r.sg[e] = ebase[offN + px];
                    const unsigned bit = (pb + e < Npix && mbase[offN + px] != 0) ? (1u << (8 * (e & 3))) : 0u;
                    if (e < 4) r.m0 |= bit;
                    else r.m1 |= bit;
                }
            }
            if (BLUE) {
                if (pb + 7 < Nb) {
                    const f4u vz0 = *reinterpret_cast<const f4u *>(zbase + offB + pb);
                    const f4u vz1 = *reinterpret_cast<const f4u *>(zbase + offB + pb + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { r.z[e] = vz0.v[e]; r.z[4 + e] = vz1.v[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) r.z[e] = zbase[offB + min(pb + e, Nb - 1)];
                }
            }
        };

        auto compute = [&](int tg, const SpecRegsX &cur, const unsigned char *tile) {
            // ---- per-element weights on the VALU (QFA/model.py:125-131)
            const float *pp = reinterpret_cast<const float *>(tile + X::OFF_PSI) + 8 * g;
            float psi[8], om[8];
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
            float c2[8], c3[8], cb[8], cb2[8];
            float qd8 = 0.f, ld8 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int px = 32 * tg + 8 * g + e;
                const unsigned mb = ((e < 4 ? cur.m0 : cur.m1) >> (8 * (e & 3))) & 0xffu;
                const bool w = svalid & (mb != 0);
                float d = cur.d[e];
                const float sg = cur.sg[e];
                float D, wD;
                if (BLUE) {
                    const bool blue = px < Nb;
                    const BlueTerms t = blue_terms(cur.z[e], k);
                    float Ab = t.A;
                    if (abase) Ab = abase[offB + min(px, Nb - 1)];        // custom tau callable (rare path)
                    const float A = blue ? Ab : 1.f;
                    const float zdom = blue ? t.zd * om[e] : 0.f;
                    D = A * A * psi[e] + zdom + sg * sg;
                    if (PREDICT) d = d - mu[min(px, Npix - 1)] * A;      // QFA/model.py:166
                    wD = w ? fast_rcp(D) : 0.f;
                    d = w ? d : 0.f;
                    const float wDA = wD * A;
                    c2[e] = wDA * A;
                    c3[e] = c2[e] * A;
                    cb[e] = wDA * d;
                    cb2[e] = c2[e] * d;
                    cblue += (w & blue) ? 1.f : 0.f;
                } else {                                                 // red side: A = 1, no omega term
                    D = psi[e] + sg * sg;
                    if (PREDICT) d = d - mu[min(px, Npix - 1)];
                    wD = w ? fast_rcp(D) : 0.f;
                    d = w ? d : 0.f;
                    c2[e] = wD;
                    cb[e] = wD * d;
                }
                qd8 += wD * d * d;
                ld8 += w ? fast_log(D) : 0.f;
                cn += w ? 1.f : 0.f;
            }
            qd += (double)qd8;
            ld += (double)ld8;
            // ---- the weights as bf16 pieces: A operands of the four contractions
            u32x4 w1h, w1m, w1l, w2h, w2m, w2l, w3h, w3m, w3l, w4h, w4m, w4l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h, m, l;
                split2(c2[2 * q], c2[2 * q + 1], h, m, l);
                w1h[q] = h; w1m[q] = m; w1l[q] = l;
                split2(cb[2 * q], cb[2 * q + 1], h, m, l);
                w3h[q] = h; w3m[q] = m; w3l[q] = l;
                if (BLUE) {
                    split2(c3[2 * q], c3[2 * q + 1], h, m, l);
                    w2h[q] = h; w2m[q] = m; w2l[q] = l;
                    split2(cb2[2 * q], cb2[2 * q + 1], h, m, l);
                    w4h[q] = h; w4m[q] = m; w4l[q] = l;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- XDL: per 16-column tile of the image three ds_read_b128, then 6 (red) or 12 (blue) MFMAs
            const unsigned char *bcol = tile + lane * 16;            // lane-linear: conflict-free ds_read_b128
            auto rdB = [&](int piece, int ct) {
                return *reinterpret_cast<const u32x4 *>(bcol + piece * X::PSTR + ct * 1024);
            };
#pragma unroll
            for (int t = 0; t < C::NFT; ++t) {
                const u32x4 bh = rdB(0, t), bm = rdB(1, t), bl = rdB(2, t);
                accb[t] = xdl6(w3h, w3m, w3l, bh, bm, bl, accb[t]);
                if (BLUE) accb2[t] = xdl6(w4h, w4m, w4l, bh, bm, bl, accb2[t]);
            }
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                const u32x4 bh = rdB(0, C::NFT + t), bm = rdB(1, C::NFT + t), bl = rdB(2, C::NFT + t);
                accC[t] = xdl6(w1h, w1m, w1l, bh, bm, bl, accC[t]);
                if (BLUE) accT[t] = xdl6(w2h, w2m, w2l, bh, bm, bl, accT[t]);
            }
        };

        // one tile: start the LDS-DMA of image tile c+1 and the spectra loads of tile c+1, compute tile c from
        // LDS buffer `buf`, ONE barrier (it also retires the DMA: the compiler waits vmcnt(0) in front of it).
        auto step = [&](int c, const SpecRegsX &cur, SpecRegsX &nxt, int buf) {
            const bool more = c + 1 < n;
            if (more) {
                stage(ta + c + 1, buf ^ 1);
                if (active) load_spec(ta + c + 1, nxt);
            }
            if (active) compute(ta + c, cur, lds[buf]);
            __syncthreads();
        };

        SpecRegsX ra, rb;
        stage(ta, 0);
        if (active) load_spec(ta, ra);
        __syncthreads();
        for (int c = 0; c < n; c += 2) {
            step(c, ra, rb, 0);
            if (c + 1 < n) step(c + 1, rb, ra, 1);
        }
    };
    run(std::false_type{}, max(t0, nbt), t1);
#pragma unroll
    for (int t = 0; t < C::NT; ++t) accT[t] = accC[t];
#pragma unroll
    for (int t = 0; t < C::NFT; ++t) accb2[t] = accb[t];
    run(std::true_type{}, t0, min(t1, nbt));

    if (!active) return;
    // C/D layout: col = lane&15, row = 4*(lane>>4) + r  -> spectrum s0 + 4g + r, column 16t + sl
    float *momseg = MOM + (size_t)seg * Bpad * C::NMOM;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ss = s0 + 4 * g + r;
        if (ss < B) {
            float *m = momseg + (size_t)ss * C::NMOM + sl;
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                m[16 * t] = accC[t][r];
                m[C::MOM_T + 16 * t] = accT[t][r];
            }
#pragma unroll
            for (int t = 0; t < C::NFT; ++t) {
                m[C::MOM_B + 16 * t] = accb[t][r];
                m[C::MOM_B2 + 16 * t] = accb2[t][r];
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
