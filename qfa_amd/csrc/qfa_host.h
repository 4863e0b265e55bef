// qfa_host.h -- host side of the step: workspace layout, launch geometry, kernel sequences.
// Shared by qfa_capi.hip (N_h <= 16; built with -amdgpu-mfma-vgpr-form so that the MFMA accumulators
// stay in VGPRs; pass 1 and stage 3 of pass 2 on the bf16 XDL pipe) and qfa_k32.hip (N_h in 17..32: 280 accumulator registers per lane need the AGPR
// half of the register file, so that translation unit is built without the flag).
#pragma once
#include "qfa_step_kernels.h"
#include "qfa_xdl_kernels.h"
#include "qfa_s12_x.h"

// pass 2 for N_h <= 16 with every contraction on the XDL pipe (qfa_grads_x.h: KP = 8 or 16, built in qfa_gx.hip)
size_t qfa_gx_image_bytes(int KP, int ntiles32);
void qfa_gx_launch(int KP, const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                   int ntiles32, const WorkPlan &wp, unsigned char *PGX, const float *SOL, const float *ZS, const float *ZP,
                   float *accum, float *slab, double *slabS, int slab_stride, Scal64 *sc64, unsigned flags, hipStream_t st,
                   bool prep = true);
// everything a training step derives from the parameters before pass 1, in one launch (N_h <= 16; qfa_gx.hip, k_prep_step)
void qfa_prep_step_launch(int KP, bool pixres, const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix,
                          int Nb, int Nh, int ntiles32, unsigned char *PFX, unsigned char *P2, float *ZS, float *zero,
                          size_t n_zero, hipStream_t st, unsigned *tick1);

// the pixel-resident form of the all-XDL pass 2 (qfa_grads_t.h, built in qfa_gx.hip; N_h = 9..16): QFA_F_PASS2_PIXRES
struct GtPlan;
size_t qfa_gt_state_bytes(int KP, int B);
int qfa_gt_items(int KP, int B, int Npix, int max_ranges);
int qfa_gt_ranges(int KP, int B, int Npix, int max_ranges);
void qfa_gt_prep_image(int KP, const qfa_params_t &p, const float *ZP, int Npix, int Nb, int Nh, unsigned char *PGT, hipStream_t st);
void qfa_gt_prep_state(int KP, const float *SOL, int B, int Nh, unsigned char *PST, hipStream_t st);
void qfa_gt_launch(int KP, const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                   int max_ranges, const unsigned char *PGT, const unsigned char *PST, const float *ZS, float *accum,
                   float *slab, double *slabS, int slab_stride, Scal64 *sc64, hipStream_t st, int *ranges_out);

// posterior writer for N_h <= 16 on the XDL pipe (qfa_predict_x.h, built in qfa_gx.hip)
size_t qfa_px_image_bytes(int KP, int ntiles32);
void qfa_px_launch(int KP, const float *F, const float *mu, int B, int Npix, int Nh, int ntiles32, const WorkPlan &wp,
                   unsigned char *PXI, const float *SOL, float *cont, float *unc, hipStream_t st);

namespace {

#ifndef QFA_P1_NW
#define QFA_P1_NW 4         // waves per workgroup of k_moments_x at N_h <= 16: 4 = two workgroups per CU; 8 = one workgroup per CU, two
                            // phase-shifted groups sharing the image ring (same results, measured slower: 1.44 against 1.31 ms at c3)
#endif
#ifndef QFA_P2_S12
#define QFA_P2_S12 1        // pass 2 at N_h > 16: 1 = k_s12_x + two k_grads_s3, 0 = k_grads (f32 stage 1) + one k_grads_s3
#endif

#ifndef QFA_P1_MAX_CHAIN
#define QFA_P1_MAX_CHAIN 32  // longest accumulation chain of pass 1 at N_h = 17..32, in 32-pixel tiles (make_layout_t).  Round 5: 64 -> 32 --
                             // F gradient of 4 096 c5-shape spectra against the float64 oracle 8.2e-5 -> 4.5e-5 (the chain bias of
                             // qfa_xdl_kernels.h, QFA_P1_FRESH: N_h <= 16 has fresh accumulators instead), c5 step 5.32 -> 5.41 ms
                             // (eight partial records per spectrum instead of four: k_sum_segments + 0.08 ms)
#endif

inline int kp_for(int Nh) { return Nh <= 8 ? 8 : (Nh <= 16 ? 16 : 32); }

// Compute units of the current device, queried once per device (launch geometry: resident-workgroup slots of the
// work plans).  256 on MI355X; also the answer when no device is visible (sizing calls in a CPU-only process).
inline int cu_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    int n = __atomic_load_n(&cache[dev], __ATOMIC_RELAXED);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 256;
    }
    __atomic_store_n(&cache[dev], n, __ATOMIC_RELAXED);
    return n;
}

// XCDs of the current device (each has its own L2; the hardware deals the workgroups of a launch to them round-robin), queried
// once per device: 8 on MI355X in SPX mode, fewer in the partitioned modes.  k_grads_t maps its work items so that the
// workgroups that walk the same spectra share an XCD (GtPlan).
inline int xcd_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 8; }
    int n = __atomic_load_n(&cache[dev], __ATOMIC_RELAXED);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || n < 1 || n > 64) {
        (void)hipGetLastError();
        n = 8;
    }
    __atomic_store_n(&cache[dev], n, __ATOMIC_RELAXED);
    return n;
}

// Work plan of a pass (WorkPlan, qfa_common.h): the blocks of 64 spectra that fill whole rounds of the 512
// resident-workgroup slots walk the whole pixel axis; the remaining blocks are cut into 1..8 pixel segments.  The
// plan minimises rounds x (tiles per item + prologue), the prologue of an item (operand loads, pipeline fill)
// counted as `pro` tiles; small batches (no full round) get the uniform split that fills the chip -- down to three
// tiles per item while the chip stays at most half full (the step of a small batch is latency-bound).
constexpr int kMaxSeg = 32;

inline WorkPlan plan_work(int B, int ntiles, int pro, int slots, int spb = 64, int max_chain = 0) {
    // slots = CUs x resident workgroups per CU; spb = spectra per block (64; 32 for k_grads_x)
    const int nblk = (B + spb - 1) / spb;
    // max_chain > 0: no item walks more than max_chain tiles (pass 1 at N_h = 17..32: see make_layout_t) -- every block is cut
    // into the same ceil(ntiles / max_chain) segments
    if (max_chain > 0 && ntiles > max_chain) {
        const int nn0 = (ntiles + max_chain - 1) / max_chain, st = (ntiles + nn0 - 1) / nn0;
        return WorkPlan{0, nblk, (ntiles + st - 1) / st, st};
    }
    WorkPlan best{0, nblk, 1, ntiles};
    double best_cost = 1e300;
    for (int full = (nblk / slots) * slots; full >= 0; full -= slots) {       // whole rounds kept unsegmented
        const int rem = nblk - full;
        for (int n = 1; n <= kMaxSeg; ++n) {
            const bool latency_bound = full == 0 && (long long)rem * n <= slots / 2;     // chip at most half full
            if (n > 1 && ntiles / n < (latency_bound ? 3 : 8)) break;  // keep segments >= 8 tiles (3 when latency-bound)
            if (n > 8 && !latency_bound) break;     // more than 8 segments measured slower once the chip is full
            const int st = (ntiles + n - 1) / n;
            const int nn = (ntiles + st - 1) / st;  // no empty segment
            const double rounds_rem = rem ? (double)((long long)rem * nn + slots - 1) / slots : 0.0;
            const double cost = (double)(full / slots) * (ntiles + pro) + (double)(long long)rounds_rem * (st + pro);
            if (cost < best_cost - 1e-9) {
                best_cost = cost;
                best = WorkPlan{full, rem, rem ? nn : 1, rem ? st : ntiles};
            }
            if (!rem) break;
        }
        if (full == 0) break;
    }
    return best;
}

struct Layout {
    int KP, NpixPad, ntiles, Bpad;
    int ntiles32;                                      // pass 1 on the XDL pipe walks 32-pixel tiles (N_h <= 16)
    WorkPlan wp1, wp2;                                 // work items of pass 1 / pass 2
    int spb1;                                          // spectra per block of pass 1's plan (128 for the 8-wave k_moments_x)
    WorkPlan wp2x;                                     // pass 2 on the XDL pipe (k_grads_x: 32-pixel tiles, 1 workgroup per CU)
    WorkPlan wpp;                                      // posterior writer on the XDL pipe (k_predict_x, N_h <= 16)
    size_t oPF, oPFT, oPFX, oPGX, oPXI, oMOM, oSOL, oNLL, oNBL, oRED, oBG, oZS, oZP, oPST, oISLAB, total;   // float offsets
    int bg_stride;                                     // beta / gamma hand-over of pass 2 at N_h = 17..32 ([2][Bpad][NpixPad])
};

template <int KP>
Layout make_layout_t(int B, int Npix) {
    using C = Cfg<KP>;
    Layout L;
    L.KP = KP;
    L.NpixPad = round_up(Npix, 16);
    L.ntiles = L.NpixPad / 16;
    L.Bpad = round_up(B, 16);
    L.ntiles32 = (Npix + 31) / 32;
    const int NCU = cu_count();
    L.wp2 = plan_work(B, L.ntiles, 4, NCU * (KP == 8 ? QFA_G8_OCC : (KP > 16 ? 1 : 2)));
    // pass 1 runs on the XDL pipe at every N_h (32-pixel tiles; one workgroup per CU at N_h > 16)
    L.spb1 = KP <= 16 ? 16 * QFA_P1_NW : 64;
    // Pass 1 sums C, T, b, b2 of a spectrum over the pixel axis in MFMA accumulators.  The matrix pipe aligns the 32 products
    // of an instruction with the accumulator they are added to and keeps about two bits below the accumulator's last place
    // (tools/ubench/mfma_round.hip: products of 1/16 ulp(C) vanish, of 1/4 ulp survive; the sum itself is rounded to nearest), so
    // the longer the chain, the more of each new product is cut off: the per-spectrum NLL of a launch that walks 250 tiles in one
    // chain is 3.3e-6 (rms) from the float64 oracle, of the same spectra in segmented small-batch launches 7.8e-7
    // (tools/chain_bias.py, c5's shape; 1.8e-6 against 6.3e-7 at c3's 125 tiles), and at N_h = 17..32 the F gradient of
    // 20 000 spectra came out 3.6e-4 from the oracle.  So at N_h = 17..32 no chain is longer than QFA_P1_MAX_CHAIN tiles (the
    // partial records are 4.5 KB per spectrum and segment: 90 MB per segment at c5, against 1.7 ms of pass 1).
#ifndef QFA_P1_MAX_CHAIN16
#define QFA_P1_MAX_CHAIN16 0   // experiment (round 5): the same bound at N_h <= 16; 0 = the work plan's own segmentation.  Superseded by
                               // the fresh accumulators per tile of k_moments_x (QFA_P1_FRESH): profiles/r5_ablation.txt, section 2
#endif
    L.wp1 = KP <= 16 ? (QFA_P1_NW == 8 ? plan_work(B, L.ntiles32, 1, NCU, 128) : plan_work(B, L.ntiles32, 1, 2 * NCU, 64, QFA_P1_MAX_CHAIN16))
                     : plan_work(B, L.ntiles32, 1, NCU, 64, QFA_P1_MAX_CHAIN);
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
    L.oPF = take((size_t)L.ntiles * C::TILE_PF);
    L.oPFT = take((size_t)L.ntiles * C::TILE_PFT);
    L.oPFX = 0;
    L.oPFX = take((size_t)L.ntiles32 * (XCfg<KP>::TILE_B / 4) + 2 * (C::FW + C::PW));      // + the column maxima (pfx_colmax)
    L.oPGX = 0;
    L.wp2x = WorkPlan{0, 0, 1, 0};
    if constexpr (KP == 8 || KP == 16) {
        L.oPGX = take(qfa_gx_image_bytes(KP, L.ntiles32) / 4);
        L.wp2x = plan_work(B, L.ntiles32, 3, NCU, 64);
    } else if constexpr (QFA_P2_S12 != 0) {            // N_h = 17..32: stages 1 and 2 of pass 2 by k_s12_x (qfa_s12_x.h)
        L.oPGX = take((size_t)L.ntiles32 * (S12<KP>::TILE_B / 4));
        L.wp2x = plan_work(B, L.ntiles32, 3, NCU, 64);
    }
    L.oPXI = 0;
    L.wpp = WorkPlan{0, 0, 1, 0};
    if constexpr (KP <= 16) {
        L.oPXI = take(qfa_px_image_bytes(KP, L.ntiles32) / 4);
        L.wpp = plan_work(B, L.ntiles32, 1, 2 * NCU, 64 * (KP == 16 ? QFA_PX_SPW : QFA_PX_SPW8));
    }
    // moment records: segment 0 for every row, segments 1.. for the rows of the segmented blocks only
    L.oMOM = take(((size_t)L.Bpad + (size_t)(L.wp1.nseg - 1) * (L.Bpad - (size_t)L.spb1 * (size_t)L.wp1.full)) * C::NMOM);
    L.oSOL = take((size_t)L.Bpad * C::NSOL);
    L.oNLL = take((size_t)L.Bpad);
    L.oNBL = take((size_t)L.Bpad);
    L.oZS = take(4 * (size_t)L.Bpad);                   // factored-z input form: per-spectrum factors ZS (float4 each)
    L.oZP = take(4 * (size_t)L.NpixPad);                //                        per-pixel factors ZP (blue pixels)
    L.oRED = take(2 * 2 * NRED + 2 + sizeof(Scal64) / 4);   // k_reduce_nll: 2 x NRED doubles + the ticket counter; then the
                                                            // float64 scalar-gradient sums of pass 2 (Scal64)
    L.oPST = 0;                                         // pixel-resident pass 2: the per-group state images (qfa_grads_t.h)
    if constexpr (KP == 8 || KP == 16) L.oPST = take(qfa_gt_state_bytes(KP, B) / 4);
    // pixel-resident pass 2 without a caller's slab: its per-range sums go through rows of the workspace and the fixed-order
    // reducer as well (no atomics: the default accumulation of a large batch is bit-reproducible too).  Sized for N_b = N_pix,
    // N_h = KP: rows | scalar records | float64 partial sums of the reducer
    L.oISLAB = 0;
    if constexpr (KP == 8 || KP == 16) {
        const size_t rows = (size_t)qfa_gt_ranges(KP, B, Npix, 1 << 30), items8 = 8 * (size_t)qfa_gt_items(KP, B, Npix, 1 << 30);
        const size_t nf = (size_t)Npix * KP + 4 * (size_t)Npix, stride = (nf + 3) / 4 * 4 + 64;
        L.oISLAB = take(rows * stride + (items8 * 3 * 2 + 8) + 2 * ((rows + 31) / 32) * nf + 16);
    }
    L.oBG = 0;
    L.bg_stride = round_up(Npix, 32);
    if constexpr (KP == 32) L.oBG = take(2 * (size_t)round_up(B, 64) * L.bg_stride);
    L.total = o;
    return L;
}

Layout make_layout(int B, int Npix, int Nh) {
    switch (kp_for(Nh)) {
        case 8: return make_layout_t<8>(B, Npix);
        case 16: return make_layout_t<16>(B, Npix);
        default: return make_layout_t<32>(B, Npix);
    }
}

// the batch's pointers: delta, error, mask always; the blue side needs zabs, or the factored form zq1 + pix_ratio
// (then zabs may be NULL), which does not combine with a host-supplied A_blue (include/qfa_hip.h)
// (ABI v3: rows / row_stride -- the resident, indexed input form; not combined with a host-supplied A_blue either)
inline int check_batch(const qfa_batch_t &b, int Npix, int Nb) {
    if (!b.delta || !b.error || !b.mask) return QFA_E_NULL;
    const bool fac = b.zq1 || b.pix_ratio;
    if (fac && !(b.zq1 && b.pix_ratio)) return QFA_E_NULL;
    if (fac && b.A_blue) return QFA_E_NULL;
    if (Nb > 0 && !fac && !b.zabs) return QFA_E_NULL;
    if (b.rows && b.A_blue) return QFA_E_NULL;
    if (b.row_stride != 0 && (b.row_stride < (int64_t)Npix || b.row_stride >= (1LL << 31))) return QFA_E_SIZE;
    if (!b.rows && (long long)64 * b.row_stride >= (1LL << 31)) return QFA_E_SIZE;   // batch order = storage order: 32-bit offsets
                                                                                    // inside a wave's 16 neighbouring rows
    return 0;
}
// the batch as the kernels take it: row_stride filled in
inline qfa_batch_t norm_batch(const qfa_batch_t &b, int Npix) {
    qfa_batch_t r = b;
    if (r.row_stride == 0) r.row_stride = Npix;
    return r;
}

inline int check_shape(int B, int Npix, int Nb, int Nh) {
    if (B < 1 || Npix < 1 || Nb < 0 || Nb > Npix || Nh < 1 || Nh > 32) return QFA_E_SIZE;
    if ((long long)64 * Npix >= (1LL << 31)) return QFA_E_SIZE;      // 32-bit byte offsets inside a wave's 16 rows
    // the per-pixel counts and the spectrum counts of the packed buffer are float32 sums of ones: exact up to 2^24
    // contributions, so one accumulation (launch, or all-reduced job of launches into one buffer) takes at most
    // 16 777 216 spectra; beyond that the caller finalises more often (the host code here never comes near it)
    if (B > (1 << 24)) return QFA_E_SIZE;
    return 0;
}

// Which form of pass 2 runs at N_h <= 16 (KP = 8 or 16).  Three forms: k_grads (float32-MFMA stage 1; the only one for
// N_h = 17..32), the two-role all-XDL k_grads_x, the pixel-resident all-XDL k_grads_t.  Defaults from measurements on MI355X
// (tools/time_pass2.py: pass 1 + solve + pass 2 per call, ms; profiles/r3_ablation_pass2.txt):
//   k_grads never: k_grads_x is as fast or faster at every batch size measured, N_h = 8: 0.072 / 0.083 at 500 spectra x 2000 px,
//     0.104 / 0.117 at 2 000, 0.192 / 0.220 at 8 000, 2.82 / 3.25 at 40 000 x 9243 (the one exception, 10 000 x 2000 --
//     157 blocks on 256 CUs, 0.281 / 0.258 -- goes to k_grads_t); N_h = 16: 2.5 - 2.6 against 3.2 ms at c3;
//   k_grads_t once the batch gives every workgroup a walk long enough to pay for its prologue and epilogue (the figures of ROUND 3 --
//     superseded for N_pix >= 1024 by the round-5 sweep quoted in pass2_use_pixres below):
//     N_h = 9..16 from 96 spectra per CU (24 576) on: N_pix = 4000: 0.146 / 0.166 at 1 000 spectra, 0.42 / 0.44 at 8 000,
//       1.43 / 1.35 at 32 000, 4.31 / 3.98 at 100 000 (k_grads_x / k_grads_t); N_pix = 640: 0.161 / 0.183 at 8 000, 1.085 / 0.971 at 100 000;
//     N_h <= 8 (a wave owns two 16-pixel tiles there) from 96 spectra per CU on, and from 36 per CU (9 216) on for N_pix >= 1024:
//       N_pix = 2000: 0.191 / 0.187 at 8 000, 0.278 / 0.215 at 10 000, 0.530 / 0.389 at 24 000, 1.21 / 0.85 at 64 000;
//       N_pix = 9243: 0.635 / 0.672 at 8 000, 1.83 / 1.63 at 24 000, 2.85 / 2.45 at 40 000; N_pix = 640: 0.116 / 0.141 at 8 000,
//       0.274 / 0.260 at 32 000, 0.83 / 0.62 at 100 000 (k_grads_x<8> / k_grads_t<8>).
// QFA_F_PASS2_F32 / QFA_F_PASS2_XDL / QFA_F_PASS2_PIXRES in the call's `flags` force one form (A/B timing and the cross-checks of
// the forms in tests/).
inline bool pass2_use_xdl(int KP, int B, unsigned flags) {
    (void)B;
    if (flags & QFA_F_PASS2_F32) return false;
    return KP == 16 || KP == 8;
}
inline bool pass2_use_pixres(int KP, int B, int Npix, unsigned flags) {
    if ((KP != 16 && KP != 8) || Npix < 16 || (flags & (QFA_F_PASS2_F32 | QFA_F_S3_FAST))) return false;
    if (flags & QFA_F_PASS2_PIXRES) return true;
    if (flags & QFA_F_PASS2_XDL) return false;
    const int ncu = cu_count();
    if (B >= 96 * ncu) return true;
    // Round 5, after the float16 stages (18 + 35 MFMAs per group instead of 36 + 51) and the reworked walk: the pixel-resident form
    // wins from a few hundred spectra on wherever the pixel axis gives every CU work (tools/pass2_crossover.sh,
    // profiles/r5_pass2_crossover.txt; step k_grads_x / k_grads_t in ms): N_h = 16, N_pix = 4000: 0.102 / 0.096 at 128 spectra,
    // 0.127 / 0.109 at 500, 0.206 / 0.149 at 2 000, 0.51 / 0.30 at 8 000; N_h = 8, N_pix = 2000: 0.080 / 0.088 at 256, 0.086 / 0.085
    // at 500, 0.095 / 0.083 at 1 000, 0.195 / 0.134 at 8 000; N_pix = 9243: 0.106 / 0.102 at 256, 0.182 / 0.148 at 1 000.
    // (Short pixel axes keep the rule of round 3: 96 spectra per CU.)
    if (Npix < 1024) return false;
    return KP == 16 ? B >= 128 : B >= 512;
}

// launch errors of the calls just made; with QFA_F_SYNC also the asynchronous ones (the stream is drained first)
inline int hip_status(hipStream_t st = nullptr, unsigned flags = 0) {
    if (flags & QFA_F_SYNC) {
        hipError_t s = hipStreamSynchronize(st);
        if (s != hipSuccess) { (void)hipGetLastError(); return (int)s; }
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <int KP>
void launch_prep(const qfa_params_t &p, const float4 *ZP, int Npix, int Nb, int Nh, const Layout &L, float *PF, float *PFT,
                 hipStream_t st) {
    dim3 blk(64, 4);
    k_prep_pf<KP><<<(L.NpixPad + 3) / 4, blk, 0, st>>>(p.F, p.Psi, p.omega, ZP, Npix, Nb, Nh, L.NpixPad, PF, PFT);
}

// MOM[seg 0] += MOM[seg 1..] for the rows of the segmented blocks of pass 1
template <int KP>
void sum_segments(float *MOM, const Layout &L, int B, hipStream_t st) {
    const WorkPlan &w = L.wp1;
    if (w.rem == 0 || w.nseg <= 1) return;
    const size_t row0 = (size_t)w.full * L.spb1;
    const size_t rows = (size_t)L.Bpad - row0;                  // Bpad is a multiple of 16, NMOM of 4
    const size_t n4 = rows * Cfg<KP>::NMOM / 4;
    k_sum_segments<<<(unsigned)((n4 + 255) / 256), 256, 0, st>>>(
        reinterpret_cast<float4 *>(MOM + row0 * Cfg<KP>::NMOM),
        reinterpret_cast<const float4 *>(MOM + (size_t)L.Bpad * Cfg<KP>::NMOM), w.nseg, n4);
}

// Factored-z input form (qfa_batch_t::zq1 / pix_ratio; qfa_common.h ZFac): the per-spectrum and per-pixel factor tables
// of this call, or NULL pointers when the batch carries zabs only.
struct ZTables {
    const float4 *ZS, *ZP;
};
inline ZTables launch_zfac(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Nb, const Layout &L,
                           float *ws, hipStream_t st) {
    if (!(b.zq1 && b.pix_ratio) || Nb <= 0) return ZTables{nullptr, nullptr};
    float4 *ZS = reinterpret_cast<float4 *>(ws + L.oZS), *ZP = reinterpret_cast<float4 *>(ws + L.oZP);
    k_zfac_spec<<<(B + 255) / 256, 256, 0, st>>>(b.zq1, b.rows, p, tau, B, ZS);
    k_zfac_pix<<<(Nb + 255) / 256, 256, 0, st>>>(b.pix_ratio, p, tau, Nb, ZP);
    return ZTables{ZS, ZP};
}

// pass 1 on the XDL pipe (split-bf16 operands, 32-pixel tiles) at every N_h
template <int KP, bool PREDICT>
void launch_moments(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, const float *mu, int B, int Npix,
                   int Nb, int Nh, const Layout &L, const ZTables &zt, float *ws, hipStream_t st, bool prep = true) {
    float *MOM = ws + L.oMOM;
    unsigned char *PFX = reinterpret_cast<unsigned char *>(ws + L.oPFX);
    if (prep) {
        if (XCfg<KP>::F16) {
            unsigned *cm = pfx_colmax<KP>(PFX, L.ntiles32);
            (void)hipMemsetAsync(cm, 0, sizeof(unsigned) * XCfg<KP>::NCOL, st);
            k_colmax<KP><<<L.ntiles32, 1024, 0, st>>>(p.F, p.Psi, Npix, Nh, cm);
        }
        k_prep_pfx<KP><<<L.ntiles32, 256, 0, st>>>(p.F, p.Psi, p.omega, PREDICT ? mu : nullptr, zt.ZP, -QFA_LOG2E * tau.offset, Npix, Nb, Nh, PFX);
    }
    constexpr int NW = KP <= 16 ? QFA_P1_NW : 4;      // (L.spb1 = 16 NW spectra per block)
    if (zt.ZS)
        k_moments_x<KP, PREDICT, NW, true><<<L.wp1.items(), 64 * NW, 0, st>>>(p, b, tau, mu, B, L.Bpad, Npix, Nb, L.ntiles32, L.wp1, PFX, zt.ZS, MOM);
    else
        k_moments_x<KP, PREDICT, NW, false><<<L.wp1.items(), 64 * NW, 0, st>>>(p, b, tau, mu, B, L.Bpad, Npix, Nb, L.ntiles32, L.wp1, PFX, nullptr, MOM);
}

// deterministic mode: slab = [nblk rows of det_row_stride floats | scalar sums: (max items) x 4 waves x 3 doubles |
//                             float64 partial sums of the reducer: det_chunks x NF]
// A row holds the NF = Npix Nh + 3 Npix + Nb sums of one block of 64 spectra, padded to a multiple of 4 floats (16-byte
// stores), plus 64 floats that lanes outside the arrays write to (every lane of a flushing wave stores: the counted
// wait in k_grads_x needs a constant number of requests per wave).
inline size_t det_rows_floats(int Npix, int Nb, int Nh) { return (size_t)Npix * Nh + 3 * (size_t)Npix + Nb; }
inline size_t det_row_stride(int Npix, int Nb, int Nh) { return (det_rows_floats(Npix, Nb, Nh) + 3) / 4 * 4 + 64; }
inline size_t det_rows(int B) { return (size_t)((B + 63) / 64); }
constexpr int DET_CHUNK_ROWS = QFA_DET_CHUNK_ROWS;   // rows summed by one thread of the reducer's first stage
inline size_t det_chunks(int B) { return (det_rows(B) + DET_CHUNK_ROWS - 1) / DET_CHUNK_ROWS; }
struct DetLayout {
    size_t NF, stride, oS, oPart, bytes;      // offsets in bytes
};
inline DetLayout det_layout(int B, int Npix, int Nb, int Nh) {
    const Layout L = make_layout(B, Npix, Nh);
    size_t items = (size_t)(L.wp2.items() > L.wp2x.items() ? L.wp2.items() : L.wp2x.items());
    if (L.KP == 16 || L.KP == 8) items = std::max(items, 2 * (size_t)qfa_gt_items(L.KP, B, Npix, (int)det_rows(B)));   // (8 waves per item there)
    DetLayout D;
    D.NF = det_rows_floats(Npix, Nb, Nh);
    D.stride = det_row_stride(Npix, Nb, Nh);
    D.oS = (det_rows(B) * D.stride * sizeof(float) + 15) / 16 * 16;
    D.oPart = D.oS + (items * 4 * 3 * sizeof(double) + 15) / 16 * 16;
    D.bytes = D.oPart + det_chunks(B) * D.NF * sizeof(double);
    return D;
}
inline size_t det_slab_bytes(int B, int Npix, int Nb, int Nh) { return det_layout(B, Npix, Nb, Nh).bytes; }

// the fixed-order reduction of the slab into the packed buffer: rows in chunks of DET_CHUNK_ROWS (float64 partials,
// rows in order), then the chunks in order
inline void launch_reduce_slab(const float *slab, const DetLayout &D, int B, int nitemwaves, float *accum, hipStream_t st,
                               int rows = -1) {
    const double *slabS = reinterpret_cast<const double *>(reinterpret_cast<const char *>(slab) + D.oS);
    double *part = reinterpret_cast<double *>(const_cast<char *>(reinterpret_cast<const char *>(slab)) + D.oPart);
    // rows written by the launch: one per block of 64 spectra, or (pixel-resident pass 2) one per range of spectra
    const int nblk = rows >= 0 ? rows : (int)det_rows(B), nch = (nblk + DET_CHUNK_ROWS - 1) / DET_CHUNK_ROWS;
    const unsigned gx = (unsigned)((D.NF + 255) / 256);
    k_reduce_slab_rows<<<dim3(gx, (unsigned)nch), 256, 0, st>>>(slab, nblk, D.NF, D.stride, part);
    k_reduce_slab_fin<<<gx, 256, 0, st>>>(part, slabS, nch, nitemwaves, D.NF, accum);
}

template <int KP>
int run_nll_grad(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                 float *nll, float *accum, float *ws, hipStream_t st, void *const *events, void *slabv, unsigned flags) {
    const Layout L = make_layout_t<KP>(B, Npix);
    float *slab = reinterpret_cast<float *>(slabv);
    DetLayout D{};
    if (slab) D = det_layout(B, Npix, Nb, Nh);
    double *slabS = slab ? reinterpret_cast<double *>(reinterpret_cast<char *>(slab) + D.oS) : nullptr;
    float *PF = ws + L.oPF, *PFT = ws + L.oPFT, *MOM = ws + L.oMOM, *SOL = ws + L.oSOL, *NBL = ws + L.oNBL;
    float *nllbuf = nll ? nll : ws + L.oNLL;
    const size_t accS = (size_t)Npix * Nh + 3 * (size_t)Npix + Nb;
    auto mark = [&](int i) {
        if (events && events[i]) (void)hipEventRecord((hipEvent_t)events[i], st);
    };
    bool pass2_xdl = false;
    if constexpr (KP == 8 || KP == 16) pass2_xdl = pass2_use_xdl(KP, B, flags);
#ifndef QFA_WITH_GFORM
    // the three-product form of stage 3 at N_h <= 16 (the G form of k_grads_x) is not part of the shipped library any more
    if constexpr (KP <= 16) { if (flags & QFA_F_S3_FAST) return QFA_E_FLAGS; }    // (the three-product form exists at N_h = 17..32 only)
#endif
    mark(0);
    const bool pixres = pass2_xdl && pass2_use_pixres(KP, B, Npix, flags);             // (its ragged-tile staging wants N_pix >= 4)
    const size_t n_acc = (size_t)Npix * Nh + 3 * (size_t)Npix + Nb + 8;
    ZTables zt{nullptr, nullptr};
    bool fused = false;
    if constexpr (KP == 8 || KP == 16) {
        // both passes on the XDL pipe: ONE launch builds the two images and the per-spectrum factors (and zeroes accum)
        if (pass2_xdl) {
            fused = true;
            const bool zf = b.zq1 && b.pix_ratio && Nb > 0;
            if (zf) zt.ZS = reinterpret_cast<float4 *>(ws + L.oZS);
            qfa_prep_step_launch(KP, pixres, p, b, tau, B, Npix, Nb, Nh, L.ntiles32, reinterpret_cast<unsigned char *>(ws + L.oPFX),
                                 reinterpret_cast<unsigned char *>(ws + L.oPGX), zf ? ws + L.oZS : nullptr,
                                 (flags & QFA_F_ZERO_ACCUM) ? accum : nullptr, n_acc, st,
                                 reinterpret_cast<unsigned *>(reinterpret_cast<double *>(ws + L.oRED) + 2 * NRED) + 1);
        }
    }
    if (!fused) {
        if (flags & QFA_F_ZERO_ACCUM) (void)hipMemsetAsync(accum, 0, n_acc * sizeof(float), st);
        zt = launch_zfac(p, b, tau, B, Nb, L, ws, st);
        // the float32 images PF / PFT serve k_grads (and the N_h = 17..32 kernels)
        launch_prep<KP>(p, zt.ZP, Npix, Nb, Nh, L, PF, PFT, st);
    }
    mark(1);
    launch_moments<KP, false>(p, b, tau, nullptr, B, Npix, Nb, Nh, L, zt, ws, st, !fused);
    mark(2);
    sum_segments<KP>(MOM, L, B, st);
    constexpr int G = 64 / KP;
    double *red = reinterpret_cast<double *>(ws + L.oRED);
    unsigned *ticket = reinterpret_cast<unsigned *>(red + 2 * NRED);
    Scal64 *sc64 = reinterpret_cast<Scal64 *>(ticket + 2);      // (zeroed by k_solve, like the ticket)
    // (pixel-resident pass 2: the solve writes the operand images that form streams instead of the float32 record SOL)
    bool solved = false;
    if constexpr (KP == 8 || KP == 16) {
        if (pixres) {
            k_solve<KP, false, true><<<(B + 4 * G - 1) / (4 * G), 256, 0, st>>>(MOM, SOL, nllbuf, NBL, B, Nh, nullptr, nullptr, ticket,
                                                                               reinterpret_cast<unsigned char *>(ws + L.oPST));
            solved = true;
        }
    }
    // small batches (one block of k_reduce_nll): the solve's last block sums the NLL itself (ticket[1]: zeroed by k_prep_step)
    bool nll_done = false;
    if constexpr (KP == 8 || KP == 16) {
        if (!solved && fused && B <= 2048) {
            k_solve<KP, false, false, true><<<(B + 4 * G - 1) / (4 * G), 256, 0, st>>>(MOM, SOL, nllbuf, NBL, B, Nh, nullptr, nullptr, ticket,
                                                                                      nullptr, accum + accS);
            solved = nll_done = true;
        }
    }
    if (!solved) k_solve<KP, false><<<(B + 4 * G - 1) / (4 * G), 256, 0, st>>>(MOM, SOL, nllbuf, NBL, B, Nh, nullptr, nullptr, ticket);
    const int nred = B <= 2048 ? 1 : (B >= 2048 * NRED ? NRED : (B + 2047) / 2048);     // small batches: one block, no hand-over
    if (!nll_done) k_reduce_nll<<<nred, 256, 0, st>>>(nllbuf, NBL, B, accum + accS, red, ticket);
    mark(3);
    if (pixres) {
        int ranges = 0;
        // the sums of a range leave as one slab row: the caller's slab (deterministic mode), else rows of the workspace
        float *rowsb = slab;
        DetLayout Dr = D;
        const int maxr = slab ? (int)det_rows(B) : (1 << 30);
        if (!slab) {
            const size_t rows = (size_t)qfa_gt_ranges(KP, B, Npix, maxr), items8 = 8 * (size_t)qfa_gt_items(KP, B, Npix, maxr);
            rowsb = ws + L.oISLAB;
            Dr.NF = det_rows_floats(Npix, Nb, Nh);
            Dr.stride = det_row_stride(Npix, Nb, Nh);
            Dr.oS = (rows * Dr.stride * sizeof(float) + 15) / 16 * 16;
            Dr.oPart = Dr.oS + (items8 * 3 * sizeof(double) + 15) / 16 * 16;
            Dr.bytes = 0;
        }
        double *rowsS = reinterpret_cast<double *>(reinterpret_cast<char *>(rowsb) + Dr.oS);
        qfa_gt_launch(KP, p, b, tau, B, Npix, Nb, Nh, maxr, reinterpret_cast<unsigned char *>(ws + L.oPGX),
                      reinterpret_cast<unsigned char *>(ws + L.oPST), reinterpret_cast<const float *>(zt.ZS), accum, rowsb, rowsS,
                      (int)Dr.stride, sc64, st, &ranges);
        launch_reduce_slab(rowsb, Dr, B, qfa_gt_items(KP, B, Npix, maxr) * 8, accum, st, ranges);
        mark(4);
        return hip_status(st, flags);
    }
    if (pass2_xdl) {
        qfa_gx_launch(KP, p, b, tau, B, Npix, Nb, Nh, L.ntiles32, L.wp2x, reinterpret_cast<unsigned char *>(ws + L.oPGX), SOL,
                      reinterpret_cast<const float *>(zt.ZS), reinterpret_cast<const float *>(zt.ZP), accum, slab, slabS,
                      (int)D.stride, sc64, flags, st, !fused);
        if (slab) launch_reduce_slab(slab, D, B, L.wp2x.items() * 4, accum, st);
        mark(4);
        return hip_status(st, flags);
    }
    if constexpr (KP == 32 && QFA_P2_S12 != 0) {
        // stages 1 and 2 for every (spectrum, pixel) on the XDL pipe, beta / gamma through HBM, then stage 3 per 16 columns
        float *BG = ws + L.oBG, *GG = BG + (size_t)round_up(B, 64) * L.bg_stride;
        unsigned char *IMG = reinterpret_cast<unsigned char *>(ws + L.oPGX);
        k_prep_s12<KP><<<L.ntiles32, 256, 0, st>>>(p.F, p.Psi, p.omega, zt.ZP, Npix, Nb, Nh, IMG);
        if (b.A_blue)
            k_s12_x<KP, true, false><<<L.wp2x.items(), 256, 0, st>>>(p, b, tau, B, Npix, Nb, Nh, L.ntiles32, L.wp2x, IMG, SOL, BG, GG, L.bg_stride, accum, slab, slabS, (int)D.stride, sc64, nullptr);
        else if (zt.ZS)
            k_s12_x<KP, false, true><<<L.wp2x.items(), 256, 0, st>>>(p, b, tau, B, Npix, Nb, Nh, L.ntiles32, L.wp2x, IMG, SOL, BG, GG, L.bg_stride, accum, slab, slabS, (int)D.stride, sc64, zt.ZS);
        else
            k_s12_x<KP, false, false><<<L.wp2x.items(), 256, 0, st>>>(p, b, tau, B, Npix, Nb, Nh, L.ntiles32, L.wp2x, IMG, SOL, BG, GG, L.bg_stride, accum, slab, slabS, (int)D.stride, sc64, nullptr);
        for (int bh = 0; 16 * bh < Nh; ++bh) {
            if (flags & QFA_F_S3_FAST)
                k_grads_s3<KP, 3><<<L.wp2.items(), 256, 0, st>>>(B, Npix, Nh, L.ntiles, L.wp2, bh, PFT, SOL, BG, GG, L.bg_stride, accum, slab, (int)D.stride);
            else
                k_grads_s3<KP, 6><<<L.wp2.items(), 256, 0, st>>>(B, Npix, Nh, L.ntiles, L.wp2, bh, PFT, SOL, BG, GG, L.bg_stride, accum, slab, (int)D.stride);
        }
        if (slab) launch_reduce_slab(slab, D, B, L.wp2x.items() * 4, accum, st);
        mark(4);
        return hip_status(st, flags);
    }
    auto grads = [&](int bh, float *BG, float *GG) {       // the float32-MFMA form: custom tau table / factored z / zabs
        if (b.A_blue)
            k_grads<KP, true, false><<<L.wp2.items(), 256, 0, st>>>(p, b, tau, B, Npix, Nb, Nh, L.ntiles, L.wp2, bh, PFT, SOL, accum, slab, slabS, (int)D.stride, sc64, nullptr, BG, GG, L.bg_stride);
        else if (zt.ZS)
            k_grads<KP, false, true><<<L.wp2.items(), 256, 0, st>>>(p, b, tau, B, Npix, Nb, Nh, L.ntiles, L.wp2, bh, PFT, SOL, accum, slab, slabS, (int)D.stride, sc64, zt.ZS, BG, GG, L.bg_stride);
        else
            k_grads<KP, false, false><<<L.wp2.items(), 256, 0, st>>>(p, b, tau, B, Npix, Nb, Nh, L.ntiles, L.wp2, bh, PFT, SOL, accum, slab, slabS, (int)D.stride, sc64, nullptr, BG, GG, L.bg_stride);
    };
    if constexpr (KP == 32) {
        // columns 0..15 by k_grads, which also stores beta and gamma; columns 16..31 by the stage-3-only kernel
        float *BG = Nh > 16 ? ws + L.oBG : nullptr, *GG = Nh > 16 ? BG + (size_t)round_up(B, 64) * L.bg_stride : nullptr;
        grads(0, BG, GG);
        if (Nh > 16)
            k_grads_s3<KP, 6><<<L.wp2.items(), 256, 0, st>>>(B, Npix, Nh, L.ntiles, L.wp2, 1, PFT, SOL, BG, GG, L.bg_stride, accum, slab, (int)D.stride);
    } else {
    for (int bh = 0; bh < (KP + 15) / 16; ++bh) {          // one launch per 16 columns of the F gradient
        if (16 * bh >= Nh) break;
        grads(bh, nullptr, nullptr);
    }
    }
    if (slab) launch_reduce_slab(slab, D, B, L.wp2.items() * 4, accum, st);
    mark(4);
    return hip_status(st, flags);
}

template <int KP>
int run_predict(const qfa_params_t &p, const float *mu, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix,
                int Nb, int Nh, float *ll, float *hmean, float *hcov, float *cont, float *unc, float *ws,
                hipStream_t st, void *const *events, unsigned flags) {
    const Layout L = make_layout_t<KP>(B, Npix);
    float *PF = ws + L.oPF, *PFT = ws + L.oPFT, *MOM = ws + L.oMOM, *SOL = ws + L.oSOL;
    auto mark = [&](int i) {
        if (events && events[i]) (void)hipEventRecord((hipEvent_t)events[i], st);
    };
    bool writer_xdl = false;
    if constexpr (KP <= 16 || QFA_P2_S12 != 0) {
        writer_xdl = !(flags & QFA_F_PREDICT_F32);               // (the float32-MFMA writer: A/B timing, cross-check)
    }
    mark(0);
    const ZTables zt = launch_zfac(p, b, tau, B, Nb, L, ws, st);
    // (PF / PFT: k_predict_out)
    if (!writer_xdl) launch_prep<KP>(p, nullptr, Npix, Nb, Nh, L, PF, PFT, st);
    launch_moments<KP, true>(p, b, tau, mu, B, Npix, Nb, Nh, L, zt, ws, st);
    mark(1);
    sum_segments<KP>(MOM, L, B, st);
    constexpr int G = 64 / KP;
    k_solve<KP, true><<<(B + 4 * G - 1) / (4 * G), 256, 0, st>>>(MOM, SOL, ll, nullptr, B, Nh, hmean, hcov);
    mark(2);
    if constexpr (KP > 16 && QFA_P2_S12 != 0) {
        if (writer_xdl) {              // the image of k_s12_x with mu in the place of Psi (qfa_s12_x.h)
            unsigned char *IMG = reinterpret_cast<unsigned char *>(ws + L.oPGX);
            k_prep_s12<KP><<<L.ntiles32, 256, 0, st>>>(p.F, mu, p.omega, nullptr, Npix, Nb, Nh, IMG);
            k_predict_x32<KP><<<L.wp2x.items(), 256, 0, st>>>(B, Npix, L.ntiles32, L.wp2x, IMG, SOL, cont, unc);
        }
    }
    if (writer_xdl && KP <= 16)
        qfa_px_launch(KP, p.F, mu, B, Npix, Nh, L.ntiles32, L.wpp, reinterpret_cast<unsigned char *>(ws + L.oPXI), SOL, cont,
                      unc, st);
    else if (!writer_xdl)
        k_predict_out<KP><<<(B + 63) / 64, 256, 0, st>>>(mu, B, Npix, L.ntiles, PFT, SOL, cont, unc);
    mark(3);
    return hip_status(st, flags);
}


}  // namespace

// N_h in 17..32 (defined in qfa_k32.hip)
int qfa_k32_nll_grad(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                     float *nll, float *accum, float *ws, hipStream_t st, void *const *events, void *slab, unsigned flags);
int qfa_k32_predict(const qfa_params_t &p, const float *mu, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix,
                    int Nb, int Nh, float *ll, float *hmean, float *hcov, float *cont, float *unc, float *ws,
                    hipStream_t st, void *const *events, unsigned flags);
