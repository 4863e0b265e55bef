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
                      unsigned flags, hipStream_t st, bool prep) {
    // QFA_WITH_GFORM (a variant build, make -C qfa_amd/csrc B=build/var_gform OUT=../libqfa_gform.so EXTRA=-DQFA_WITH_GFORM=1): the three-product G form of stage 3 behind QFA_F_S3_FAST --
    // round 2's headline form, kept out of the shipped library from round 4 on (the host refuses the flag at N_h <= 16)
#ifndef QFA_WITH_GFORM
#define QFA_WITH_GFORM 0
#endif
    const bool fast = QFA_WITH_GFORM && (flags & QFA_F_S3_FAST) != 0;
    if (prep) k_prep_pgx<KP><<<ntiles32, 256, 0, st>>>(p.F, p.Psi, p.omega, ZP, Npix, Nb, Nh, fast ? 0 : 1, PGX);
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
                   float *accum, float *slab, double *slabS, int slab_stride, Scal64 *sc64, unsigned flags, hipStream_t st,
                   bool prep) {
    const float4 *zs = reinterpret_cast<const float4 *>(ZS), *zp = reinterpret_cast<const float4 *>(ZP);
    if (KP == 8) gx_launch<8>(p, b, tau, B, Npix, Nb, Nh, ntiles32, wp, PGX, SOL, zs, zp, accum, slab, slabS, slab_stride, sc64, flags, st, prep);
    else gx_launch<16>(p, b, tau, B, Npix, Nb, Nh, ntiles32, wp, PGX, SOL, zs, zp, accum, slab, slabS, slab_stride, sc64, flags, st, prep);
}

// ---- everything a training step derives from the parameters before pass 1, in ONE launch (N_h <= 16): the pass-1 image,
// the pass-2 image of the form that will run, the per-spectrum factors of the factored-z form, and (QFA_F_ZERO_ACCUM) the
// zeroing of the packed buffer.  The step of a small batch -- the reference's default is 500 spectra -- is a chain of a dozen
// dependent kernels of 3 - 30 us with ~3 us of dispatch latency between any two: five of them were these preparations
// (k_zfac_spec, k_zfac_pix, k_prep_pfx, k_prep_pgx / k_prep_pgt, torch's fill).  Block ranges: [pfx tiles | pass-2 tiles |
// 256 spectra each | 1024 floats of accum each]; the per-pixel factors are computed where they are needed (ZPSrc).
template <int KP, bool PIXRES>
__global__ __launch_bounds__(256) void k_prep_step(qfa_params_t p, qfa_tau_t tau, const float *__restrict__ zq1,
                                                   const float *__restrict__ pix_ratio, const int *__restrict__ rows, int B,
                                                   int Npix, int Nb, int Nh, int n_pfx, int n_p2, int n_zs,
                                                   unsigned char *__restrict__ PFX, unsigned char *__restrict__ P2,
                                                   float4 *__restrict__ ZS, float *__restrict__ zero, size_t n_zero,
                                                   unsigned *__restrict__ tick1) {
    const int b = (int)blockIdx.x;
    if (b == 0 && threadIdx.x == 0) *tick1 = 0u;             // arrival counter of k_solve<.., NLLRED> later in this step
    const ZPSrc zp{nullptr, pix_ratio, p.beta, tau.expo, -QFA_LOG2E * tau.offset};      // (offp as load_consts forms it)
    if (b < n_pfx) prep_pfx_body<KP>(b, n_pfx, p.F, p.Psi, p.omega, nullptr, zp, Npix, Nb, Nh, PFX);
    else if (b < n_pfx + n_p2) {
        if constexpr (PIXRES) prep_pgt_body<KP>(b - n_pfx, p.F, p.Psi, p.omega, zp, Npix, Nb, Nh, P2);
        else prep_pgx_body<KP>(b - n_pfx, p.F, p.Psi, p.omega, zp, Npix, Nb, Nh, 1, P2);
    } else if (b < n_pfx + n_p2 + n_zs) zfac_spec_body((b - n_pfx - n_p2) * 256 + (int)threadIdx.x, zq1, rows, p, tau, B, ZS);
    else {
        const size_t i0 = (size_t)(b - n_pfx - n_p2 - n_zs) * 1024 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + 256 * k < n_zero) zero[i0 + 256 * k] = 0.f;
    }
}
void qfa_prep_step_launch(int KP, bool pixres, const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix,
                          int Nb, int Nh, int ntiles32, unsigned char *PFX, unsigned char *P2, float *ZS, float *zero,
                          size_t n_zero, hipStream_t st, unsigned *tick1) {
    const bool zf = ZS != nullptr;
    const int n_pfx = ntiles32, n_zs = zf ? (B + 255) / 256 : 0;
    const int pxw = KP == 8 ? GTT<8>::PXW : GTT<16>::PXW, tpw = KP == 8 ? GTT<8>::TPW : GTT<16>::TPW;
    const int n_p2 = pixres ? (Npix + pxw - 1) / pxw * tpw : ntiles32;
    const size_t nz = zero ? n_zero : 0;
    const unsigned grid = (unsigned)(n_pfx + n_p2 + n_zs + (nz + 1023) / 1024);
    float4 *zs = reinterpret_cast<float4 *>(ZS);
    const float *zq1 = zf ? b.zq1 : nullptr, *ratio = zf ? b.pix_ratio : nullptr;
    auto go = [&](auto kp, auto px) {
        constexpr int KPc = decltype(kp)::value;
        if (XCfg<KPc>::F16) {         // the column maxima of the pass-1 image, in front of its builder
            unsigned *cm = pfx_colmax<KPc>(PFX, n_pfx);
            (void)hipMemsetAsync(cm, 0, sizeof(unsigned) * XCfg<KPc>::NCOL, st);
            k_colmax<KPc><<<n_pfx, 1024, 0, st>>>(p.F, p.Psi, Npix, Nh, cm);
        }
        k_prep_step<decltype(kp)::value, decltype(px)::value><<<grid, 256, 0, st>>>(p, tau, zq1, ratio, b.rows, B, Npix, Nb, Nh, n_pfx, n_p2,
                                                                                    n_zs, PFX, P2, zs, zero, nz, tick1);
    };
    using K8 = std::integral_constant<int, 8>;
    using K16 = std::integral_constant<int, 16>;
    if (KP == 8) { if (pixres) go(K8{}, std::true_type{}); else go(K8{}, std::false_type{}); }
    else { if (pixres) go(K16{}, std::true_type{}); else go(K16{}, std::false_type{}); }
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
