// qfa_gt.hip -- the pixel-resident form of pass 2 (qfa_grads_t.h) in its own translation unit: its register budget (72
// image + 64 accumulator registers per wave, two waves per SIMD) wants its own scheduler settings (Makefile, GTFLAGS).
#include "qfa_grads_t.h"

#include "qfa_host.h"

// ---- pass 2, pixel-resident form (qfa_grads_t.h)
size_t qfa_gt_state_bytes(int KP, int B) { return KP == 16 ? (size_t)((B + 15) / 16) * GTT<16>::STATE_B : 0; }
// tiles, pixel blocks of 8 tiles, ranges of spectra groups: about one workgroup per CU; a multiple of 8 ranges (the
// workgroups of a range then share an XCD, block = pb R + r) where the batch has the groups for it
static GtPlan gt_plan(int B, int Npix, int max_ranges) {
    GtPlan g;
    g.T16 = (Npix + 15) / 16;
    g.PB = (g.T16 + 7) / 8;
    const int G = (B + 15) / 16, ncu = cu_count();
    int R = std::max(1, (ncu + g.PB / 2) / g.PB);
    if (R >= 8) R = (R + 3) / 8 * 8;
    R = std::min(R, std::max(1, G / 2));                 // at least two groups per range
    R = std::max(1, std::min(R, max_ranges));
    g.gpr = (G + R - 1) / R;
    g.R = (G + g.gpr - 1) / g.gpr;
    return g;
}
int qfa_gt_items(int B, int Npix, int max_ranges) { return gt_plan(B, Npix, max_ranges).items(); }
// the per-tile images (beside the other parameter images of the call) and the per-group operand images (behind the solve)
void qfa_gt_prep_image(const qfa_params_t &p, const float *ZP, int Npix, int Nb, int Nh, unsigned char *PGT, hipStream_t st) {
    k_prep_pgt<16><<<(Npix + 15) / 16, 256, 0, st>>>(p.F, p.Psi, p.omega, reinterpret_cast<const float4 *>(ZP), Npix, Nb, Nh, PGT);
}
void qfa_gt_prep_state(const float *SOL, int B, int Nh, unsigned char *PST, hipStream_t st) {
    k_prep_pst<16><<<(B + 15) / 16, 256, 0, st>>>(SOL, B, Nh, PST);
}
void qfa_gt_launch(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                   int max_ranges, const unsigned char *PGT, const unsigned char *PST, const float *ZS, float *accum,
                   float *slab, double *slabS, int slab_stride, Scal64 *sc64, hipStream_t st, int *ranges_out) {
    const GtPlan g = gt_plan(B, Npix, max_ranges);
    if (ranges_out) *ranges_out = g.R;
    const float4 *zs = reinterpret_cast<const float4 *>(ZS);
    auto go = [&](auto hasa, auto zf) {
        k_grads_t<16, decltype(hasa)::value, decltype(zf)::value><<<g.items(), 512, 0, st>>>(
            p, b, tau, B, Npix, Nb, Nh, g, PGT, PST, zs, accum, slab, slabS, slab_stride, sc64);
    };
    if (b.A_blue) go(std::true_type{}, std::false_type{});
    else if (ZS) go(std::false_type{}, std::true_type{});
    else go(std::false_type{}, std::false_type{});
}

#if QFA_GT_STAMPS
extern "C" int qfa_gt_debug_stamps(unsigned long long *out) {      // diagnostic build only
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qfa_gt_stamps), 32 * sizeof(unsigned long long));
}
#endif
