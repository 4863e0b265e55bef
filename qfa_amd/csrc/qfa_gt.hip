// qfa_gt.hip -- the pixel-resident form of pass 2 (qfa_grads_t.h) in its own translation unit: its register budget (72
// image + 64 accumulator registers per wave, two waves per SIMD) wants its own scheduler settings (Makefile, GTFLAGS).
#include "qfa_grads_t.h"

#include "qfa_host.h"

// ---- pass 2, pixel-resident form (qfa_grads_t.h)
size_t qfa_gt_state_bytes(int KP, int B) {
    return KP == 16 ? (size_t)((B + 15) / 16) * GTT<16>::STATE_B : (KP == 8 ? (size_t)((B + 15) / 16) * GTT<8>::STATE_B : 0);
}
// tiles, pixel blocks of 8 tiles, ranges of spectra groups: about one workgroup per CU; a multiple of 8 ranges (the
// workgroups of a range then share an XCD, block = pb R + r) where the batch has the groups for it
static GtPlan gt_plan(int KP, int B, int Npix, int max_ranges) {
    GtPlan g;
    g.nxcd = xcd_count();
    const int pxw = KP == 8 ? GTT<8>::PXW : GTT<16>::PXW;       // pixels per wave
    g.T16 = (Npix + pxw - 1) / pxw;
    g.PB = (g.T16 + 7) / 8;
    const int G = (B + 15) / 16, ncu = cu_count();
    // ranges: the workgroups run one per CU in rounds, so R is chosen for full rounds -- the smallest R whose
    // rounds(PB R) / R (time per unit of work) is within 3 % of the best; then a multiple of 8 if that costs nothing
    int R = 1;
    {
        const int rmax = std::max(1, std::min(4 * ncu / std::max(1, g.PB) + 1, std::max(1, G / 2)));
        auto cost = [&](int r) { return (double)((g.PB * r + ncu - 1) / ncu) / r; };
        double best = cost(1);
        for (int r = 2; r <= rmax; ++r) best = std::min(best, cost(r));
        for (int r = 1; r <= rmax; ++r)
            if (cost(r) <= 1.03 * best) { R = r; break; }
        const int X = g.nxcd;
        if (R >= X && R % X != 0 && (R + X - 1) / X * X <= rmax && cost((R + X - 1) / X * X) <= cost(R)) R = (R + X - 1) / X * X;
    }
    R = std::max(1, std::min(R, max_ranges));
    g.gpr = (G + R - 1) / R;
    g.R = (G + g.gpr - 1) / g.gpr;
    return g;
}
int qfa_gt_items(int KP, int B, int Npix, int max_ranges) { return gt_plan(KP, B, Npix, max_ranges).items(); }
int qfa_gt_ranges(int KP, int B, int Npix, int max_ranges) { return gt_plan(KP, B, Npix, max_ranges).R; }
// the per-tile images (beside the other parameter images of the call) and the per-group operand images (behind the solve)
void qfa_gt_prep_image(int KP, const qfa_params_t &p, const float *ZP, int Npix, int Nb, int Nh, unsigned char *PGT, hipStream_t st) {
    const float4 *zp = reinterpret_cast<const float4 *>(ZP);
    // (one block per 16-pixel tile: TPW per wave tile)
    if (KP == 8) k_prep_pgt<8><<<(Npix + GTT<8>::PXW - 1) / GTT<8>::PXW * GTT<8>::TPW, 256, 0, st>>>(p.F, p.Psi, p.omega, zp, Npix, Nb, Nh, PGT);
    else k_prep_pgt<16><<<(Npix + GTT<16>::PXW - 1) / GTT<16>::PXW * GTT<16>::TPW, 256, 0, st>>>(p.F, p.Psi, p.omega, zp, Npix, Nb, Nh, PGT);
}
void qfa_gt_prep_state(int KP, const float *SOL, int B, int Nh, unsigned char *PST, hipStream_t st) {
    if (KP == 8) k_prep_pst<8><<<(B + 15) / 16, 256, 0, st>>>(SOL, B, Nh, PST);
    else k_prep_pst<16><<<(B + 15) / 16, 256, 0, st>>>(SOL, B, Nh, PST);
}
template <int KP>
static void gt_launch(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                      const GtPlan &g, const unsigned char *PGT, const unsigned char *PST, const float4 *zs, float *accum,
                      float *slab, double *slabS, int slab_stride, Scal64 *sc64, hipStream_t st) {
    auto go = [&](auto hasa, auto zf, auto ix) {
        k_grads_t<KP, decltype(hasa)::value, decltype(zf)::value, decltype(ix)::value><<<g.items(), 512, 0, st>>>(
            p, b, tau, B, Npix, Nb, Nh, g, PGT, PST, zs, accum, slab, slabS, slab_stride, sc64);
    };
    using T = std::true_type;
    using F = std::false_type;
    if (b.A_blue) go(T{}, F{}, F{});                         // (not combined with rows: check_batch)
    else if (zs) { if (b.rows) go(F{}, T{}, T{}); else go(F{}, T{}, F{}); }
    else { if (b.rows) go(F{}, F{}, T{}); else go(F{}, F{}, F{}); }
}
void qfa_gt_launch(int KP, const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                   int max_ranges, const unsigned char *PGT, const unsigned char *PST, const float *ZS, float *accum,
                   float *slab, double *slabS, int slab_stride, Scal64 *sc64, hipStream_t st, int *ranges_out) {
    const GtPlan g = gt_plan(KP, B, Npix, max_ranges);
    if (ranges_out) *ranges_out = g.R;
    const float4 *zs = reinterpret_cast<const float4 *>(ZS);
    if (KP == 8) gt_launch<8>(p, b, tau, B, Npix, Nb, Nh, g, PGT, PST, zs, accum, slab, slabS, slab_stride, sc64, st);
    else gt_launch<16>(p, b, tau, B, Npix, Nb, Nh, g, PGT, PST, zs, accum, slab, slabS, slab_stride, sc64, st);
}

#if QFA_GT_STAMPS
extern "C" int qfa_gt_debug_stamps(unsigned long long *out) {      // diagnostic build only
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qfa_gt_stamps), 32 * sizeof(unsigned long long));
}
#endif
