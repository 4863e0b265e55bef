// qfa_gx.hip -- pass 2 on the XDL pipe (qfa_grads_x.h) in its own translation unit: the kernel is large and
// is iterated on separately from the rest of the library.
#include "qfa_grads_x.h"
#include "qfa_grads_t.h"       // (GTT: the image size; the kernel itself is built in qfa_gt.hip)
#include "qfa_predict_x.h"

#include "qfa_host.h"

size_t qfa_gx_image_bytes(int KP, int ntiles32) {            // the largest of the forms' images (one region serves all)
    const size_t x = KP == 8 ? GXT<8>::TILE_B : GXT<16>::TILE_B;
    const size_t t = 2 * (size_t)(KP == 16 ? GTT<16>::TILE_B : GTT<8>::TILE_B);      // (two 16-pixel tiles per 32 pixels)
    return (size_t)ntiles32 * std::max(t, x);
}

// pass 2, two-role form (qfa_grads_x.h)
template <int KP>
static void gx_launch(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                      int ntiles32, const WorkPlan &wp, unsigned char *PGX, const float *SOL, const float4 *ZS,
                      const float4 *ZP, float *accum, float *slab, double *slabS, int slab_stride, Scal64 *sc64,
                      unsigned flags, hipStream_t st) {
    // QFA_WITH_GFORM (a variant build, tools/build_full_variant.sh gform -DQFA_WITH_GFORM=1): the three-product G form of stage 3 behind QFA_F_S3_FAST --
    // round 2's headline form, kept out of the shipped library from round 4 on (the host refuses the flag at N_h <= 16)
#ifndef QFA_WITH_GFORM
#define QFA_WITH_GFORM 0
#endif
    const bool fast = QFA_WITH_GFORM && (flags & QFA_F_S3_FAST) != 0;
    k_prep_pgx<KP><<<ntiles32, 256, 0, st>>>(p.F, p.Psi, p.omega, ZP, Npix, Nb, Nh, fast ? 0 : 1, PGX);
    auto go = [&](auto hasa, auto terms, auto zf) {
        k_grads_x<KP, decltype(hasa)::value, decltype(terms)::value, decltype(zf)::value><<<wp.items(), 512, 0, st>>>(
            p, b, tau, B, Npix, Nb, Nh, ntiles32, wp, PGX, SOL, ZS, accum, slab, slabS, slab_stride, sc64);
    };
    using T6 = std::integral_constant<int, 6>;
#if QFA_WITH_GFORM
    using T3 = std::integral_constant<int, 3>;
    if (fast) {
        if (b.A_blue) go(std::true_type{}, T3{}, std::false_type{});
        else if (ZS) go(std::false_type{}, T3{}, std::true_type{});
        else go(std::false_type{}, T3{}, std::false_type{});
        return;
    }
#endif
    if (b.A_blue) go(std::true_type{}, T6{}, std::false_type{});
    else if (ZS) go(std::false_type{}, T6{}, std::true_type{});
    else go(std::false_type{}, T6{}, std::false_type{});
}
void qfa_gx_launch(int KP, const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                   int ntiles32, const WorkPlan &wp, unsigned char *PGX, const float *SOL, const float *ZS, const float *ZP,
                   float *accum, float *slab, double *slabS, int slab_stride, Scal64 *sc64, unsigned flags, hipStream_t st) {
    const float4 *zs = reinterpret_cast<const float4 *>(ZS), *zp = reinterpret_cast<const float4 *>(ZP);
    if (KP == 8) gx_launch<8>(p, b, tau, B, Npix, Nb, Nh, ntiles32, wp, PGX, SOL, zs, zp, accum, slab, slabS, slab_stride, sc64, flags, st);
    else gx_launch<16>(p, b, tau, B, Npix, Nb, Nh, ntiles32, wp, PGX, SOL, zs, zp, accum, slab, slabS, slab_stride, sc64, flags, st);
}

size_t qfa_px_image_bytes(int KP, int ntiles32) {
    return (size_t)ntiles32 * (KP == 8 ? PX<8>::TILE_B : PX<16>::TILE_B);
}

void qfa_px_launch(int KP, const float *F, const float *mu, int B, int Npix, int Nh, int ntiles32, const WorkPlan &wp,
                   unsigned char *PXI, const float *SOL, float *cont, float *unc, hipStream_t st) {
    if (KP == 8) {
        k_prep_px<8><<<ntiles32, 256, 0, st>>>(F, Npix, Nh, PXI);
        if (QFA_PX_SPW8 == 1 && px_realign(KP, Npix, cont, unc))
            k_predict_x<8, 1, true><<<wp.items(), 256, 0, st>>>(mu, B, Npix, ntiles32, wp, PXI, SOL, cont, unc);
        else
            k_predict_x<8, QFA_PX_SPW8><<<wp.items(), 256, 0, st>>>(mu, B, Npix, ntiles32, wp, PXI, SOL, cont, unc);
    } else {
        k_prep_px<16><<<ntiles32, 256, 0, st>>>(F, Npix, Nh, PXI);
        k_predict_x<16, QFA_PX_SPW><<<wp.items(), 256, 0, st>>>(mu, B, Npix, ntiles32, wp, PXI, SOL, cont, unc);
    }
}

#if QFA_GX_STAMPS
extern "C" int qfa_gx_debug_stamps(unsigned long long *out) {      // diagnostic build only
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qfa_gx_stamps), 64 * sizeof(unsigned long long));
}
#endif
