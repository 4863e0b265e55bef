// qfa_capi.hip -- C-ABI of libqfa_hip.so (include/qfa_hip.h): argument checks, workspace layout,
// launch geometry.  Kernels live in qfa_step_kernels.h (hot path) and qfa_small_kernels.h.
#include "qfa_host.h"
#include "qfa_small_kernels.h"
#include "qfa_prep_kernels.h"

// ================================================================================================
// C ABI
// ================================================================================================
namespace {

const double kLymanF[30] = {4.1620e-01, 7.9140e-02, 2.9010e-02, 1.3950e-02, 7.8030e-03, 4.8160e-03, 3.1850e-03,
                            2.2170e-03, 1.6060e-03, 1.2010e-03, 9.2190e-04, 7.2310e-04, 5.7770e-04, 4.6890e-04,
                            3.8580e-04, 3.2120e-04, 2.7030e-04, 2.2970e-04, 1.9680e-04, 1.6990e-04, 1.4770e-04,
                            1.2930e-04, 1.1370e-04, 1.0060e-04, 8.9360e-05, 7.9780e-05, 7.1480e-05, 6.4350e-05,
                            5.8120e-05, 5.2640e-05};
const double kLymanLam[30] = {1215.6701, 1025.7222, 972.5367, 949.7430, 937.8034, 930.7482, 926.2256, 923.1503,
                              920.9630,  919.3513,  918.1293, 917.1805, 916.4291, 915.8238, 915.3289, 914.9192,
                              914.5762,  914.2861,  914.0385, 913.8256, 913.6411, 913.4803, 913.3391, 913.2146,
                              913.1042,  913.0059,  912.9179, 912.8389, 912.7676, 912.7032};

}  // namespace

extern "C" {

#if QFA_P1_STAMPS
extern "C" int qfa_p1_debug_stamps(unsigned long long *out) {   // diagnostic build only (tools/p1_stamps.sh)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qfa_p1_stamps), 16 * sizeof(unsigned long long));
}
#endif
#if QFA_ABL == 7
int qfa_debug_stamps(unsigned long long *out) {      // diagnostic build only
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qfa_dbg_stamps), 64 * sizeof(unsigned long long));
}
#endif

int qfa_abi_version(void) { return QFA_ABI_VERSION; }

int qfa_tau_model(int which, int series, qfa_tau_t *out) {
    if (!out) return QFA_E_NULL;
    if (series < 1 || series > 30) return QFA_E_TAU;
    const double coeff = kLymanLam[series - 1] * kLymanF[series - 1] / (kLymanLam[0] * kLymanF[0]);
    double amp, scale, expo, off;
    switch (which) {
        case QFA_TAU_BECKER: amp = 0.751; scale = 1.0 / 4.5; expo = 2.90; off = -0.132; break;
        case QFA_TAU_FG: amp = 0.0018; scale = 1.0; expo = 3.92; off = 0.0; break;
        case QFA_TAU_KAMBLE: amp = 5.54e-3; scale = 1.0; expo = 3.182; off = 0.0; break;
        case QFA_TAU_MOCK: amp = 0.2231435513142097; scale = 1.0 / 3.25; expo = 3.2; off = 0.0; break;
        default: return QFA_E_TAU;
    }
    out->amp = (float)(amp * coeff);
    out->scale = (float)scale;
    out->expo = (float)expo;
    out->offset = (float)(off * coeff);
    return 0;
}

size_t qfa_workspace_bytes(int B, int Npix, int Nh) {
    if (B < 1 || Npix < 1 || Nh < 1 || Nh > 32) return 0;
    return make_layout(B, Npix, Nh).total * sizeof(float);
}

size_t qfa_accum_floats(int Npix, int Nb, int Nh) {
    return (size_t)Npix * Nh + 3 * (size_t)Npix + (size_t)Nb + 8;
}

int qfa_nll_grad_events_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau, int B, int Npix,
                            int Nb, int Nh, float *nll, float *accum, void *workspace, size_t workspace_bytes,
                            void *stream, void *const *events);

int qfa_nll_grad_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau, int B, int Npix, int Nb,
                     int Nh, float *nll, float *accum, void *workspace, size_t workspace_bytes, void *stream) {
    return qfa_nll_grad_events_f32(p, b, tau, B, Npix, Nb, Nh, nll, accum, workspace, workspace_bytes, stream, nullptr);
}

int qfa_nll_grad_events_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau, int B, int Npix,
                            int Nb, int Nh, float *nll, float *accum, void *workspace, size_t workspace_bytes,
                            void *stream, void *const *events) {
    return qfa_nll_grad_det_f32(p, b, tau, B, Npix, Nb, Nh, nll, accum, workspace, workspace_bytes, nullptr, 0, stream,
                                events);
}

size_t qfa_det_slab_bytes(int B, int Npix, int Nb, int Nh) {
    if (B < 1 || Npix < 1 || Nb < 0 || Nb > Npix || Nh < 1 || Nh > 32) return 0;
    return det_slab_bytes(B, Npix, Nb, Nh);
}

int qfa_nll_grad_det_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau, int B, int Npix, int Nb,
                         int Nh, float *nll, float *accum, void *workspace, size_t workspace_bytes, void *slab,
                         size_t slab_bytes, void *stream, void *const *events) {
    return qfa_nll_grad_ex_f32(p, b, tau, B, Npix, Nb, Nh, nll, accum, workspace, workspace_bytes, slab, slab_bytes, 0u,
                               stream, events);
}

int qfa_nll_grad_ex_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau, int B, int Npix, int Nb,
                        int Nh, float *nll, float *accum, void *workspace, size_t workspace_bytes, void *slab,
                        size_t slab_bytes, unsigned flags, void *stream, void *const *events) {
    if (!p || !b || !tau || !accum || !workspace) return QFA_E_NULL;
    if (!p->F || !p->Psi || !p->tau0 || !p->c0 || !p->beta || (Nb > 0 && !p->omega)) return QFA_E_NULL;
    if (int e = check_batch(*b, Npix, Nb)) return e;
    if (int e = check_shape(B, Npix, Nb, Nh)) return e;
    if (workspace_bytes < qfa_workspace_bytes(B, Npix, Nh)) return QFA_E_WORKSPACE;
    if (slab && slab_bytes < det_slab_bytes(B, Npix, Nb, Nh)) return QFA_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const qfa_batch_t bb = norm_batch(*b, Npix);
    if (kp_for(Nh) == 8) return run_nll_grad<8>(*p, bb, *tau, B, Npix, Nb, Nh, nll, accum, ws, st, events, slab, flags);
    if (kp_for(Nh) == 16) return run_nll_grad<16>(*p, bb, *tau, B, Npix, Nb, Nh, nll, accum, ws, st, events, slab, flags);
    return qfa_k32_nll_grad(*p, bb, *tau, B, Npix, Nb, Nh, nll, accum, ws, st, events, slab, flags);
}

int qfa_finalize_grads_f32(const float *accum, const float *F, int Npix, int Nb, int Nh, int normalize, float *gF,
                           float *gPsi, float *gOmega, float *gTau0, float *gC0, float *gBeta, float *loss,
                           void *stream) {
    if (!accum || !F || !gF || !gPsi || !gTau0 || !gC0 || !gBeta || !loss || (Nb > 0 && !gOmega)) return QFA_E_NULL;
    if (int e = check_shape(1, Npix, Nb, Nh)) return e;
    const size_t n = (size_t)Npix * Nh;
    k_finalize<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(accum, F, Npix, Nb, Nh, normalize, gF, gPsi,
                                                                            gOmega, gTau0, gC0, gBeta, loss);
    return hip_status();
}

int qfa_predict_f32(const qfa_params_t *p, const float *mu, const qfa_batch_t *b, const qfa_tau_t *tau, int B,
                    int Npix, int Nb, int Nh, float *ll, float *hmean, float *hcov, float *cont, float *unc,
                    void *workspace, size_t workspace_bytes, void *stream) {
    return qfa_predict_events_f32(p, mu, b, tau, B, Npix, Nb, Nh, ll, hmean, hcov, cont, unc, workspace, workspace_bytes,
                                  stream, nullptr);
}

int qfa_predict_events_f32(const qfa_params_t *p, const float *mu, const qfa_batch_t *b, const qfa_tau_t *tau, int B,
                           int Npix, int Nb, int Nh, float *ll, float *hmean, float *hcov, float *cont, float *unc,
                           void *workspace, size_t workspace_bytes, void *stream, void *const *events) {
    return qfa_predict_ex_f32(p, mu, b, tau, B, Npix, Nb, Nh, ll, hmean, hcov, cont, unc, workspace, workspace_bytes, 0u,
                              stream, events);
}

int qfa_predict_ex_f32(const qfa_params_t *p, const float *mu, const qfa_batch_t *b, const qfa_tau_t *tau, int B,
                       int Npix, int Nb, int Nh, float *ll, float *hmean, float *hcov, float *cont, float *unc,
                       void *workspace, size_t workspace_bytes, unsigned flags, void *stream, void *const *events) {
    if (!p || !b || !tau || !mu || !ll || !hmean || !hcov || !cont || !unc || !workspace) return QFA_E_NULL;
    if (!p->F || !p->Psi || !p->tau0 || !p->c0 || !p->beta || (Nb > 0 && !p->omega)) return QFA_E_NULL;
    if (int e = check_batch(*b, Npix, Nb)) return e;
    if (int e = check_shape(B, Npix, Nb, Nh)) return e;
    if (workspace_bytes < qfa_workspace_bytes(B, Npix, Nh)) return QFA_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const qfa_batch_t bb = norm_batch(*b, Npix);
    if (kp_for(Nh) == 8) return run_predict<8>(*p, mu, bb, *tau, B, Npix, Nb, Nh, ll, hmean, hcov, cont, unc, ws, st, events, flags);
    if (kp_for(Nh) == 16) return run_predict<16>(*p, mu, bb, *tau, B, Npix, Nb, Nh, ll, hmean, hcov, cont, unc, ws, st, events, flags);
    return qfa_k32_predict(*p, mu, bb, *tau, B, Npix, Nb, Nh, ll, hmean, hcov, cont, unc, ws, st, events, flags);
}

int qfa_adam_clip_f32(const float *p, const float *g, float *m, float *v, float *p_out, size_t n, double lr, double b1,
                      double b2, double eps, double wd, int i, float lo, float hi, void *stream) {
    if (n == 0) return 0;
    if (!p || !g || !m || !v || !p_out) return QFA_E_NULL;
    if (i < 0) return QFA_E_SIZE;
    // The reference mixes Python floats (double) with float32 tensors: every scalar below is
    // formed in double and rounded to float32 once, exactly where torch would round it.
    const float bc1 = (float)(1.0 - pow(b1, (double)(i + 1))), bc2 = (float)(1.0 - pow(b2, (double)(i + 1)));
    k_adam_clip<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        p, g, m, v, p_out, n, (float)lr, (float)b1, (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps,
        (float)wd, bc1, bc2, lo, hi);
    return hip_status();
}

int qfa_adam_clip_multi_f32(const qfa_adam_multi_t *t, double lr, double b1, double b2, double eps, double wd, int i,
                            void *stream) {
    if (!t) return QFA_E_NULL;
    if (t->count < 0 || t->count > QFA_ADAM_MAX || i < 0) return QFA_E_SIZE;
    AdamMultiArgs a;
    a.t = *t;
    unsigned nblk = 0;
    for (int k = 0; k < t->count; ++k) {
        // an empty tensor (omega of a model without blue pixels) is a no-op, whatever its pointers are
        if (t->n[k] != 0 && (!t->p[k] || !t->g[k] || !t->m[k] || !t->v[k] || !t->p_out[k])) return QFA_E_NULL;
        a.blk0[k] = nblk;
        nblk += (unsigned)((t->n[k] + 255) / 256);
    }
    for (int k = t->count; k <= QFA_ADAM_MAX; ++k) a.blk0[k] = nblk;
    if (nblk == 0) return 0;
    const float bc1 = (float)(1.0 - pow(b1, (double)(i + 1))), bc2 = (float)(1.0 - pow(b2, (double)(i + 1)));
    k_adam_clip_multi<<<nblk, 256, 0, (hipStream_t)stream>>>(a, (float)lr, (float)b1, (float)b2, (float)(1.0 - b1),
                                                             (float)(1.0 - b2), (float)eps, (float)wd, bc1, bc2);
    return hip_status();
}

int qfa_finalize_adam_clip_f32(const float *accum, int Npix, int Nb, int Nh, const qfa_adam_multi_t *t, double lr, double b1,
                               double b2, double eps, double wd, int i, float *loss, void *stream) {
    if (!accum || !t || !loss) return QFA_E_NULL;
    if (int e = check_shape(1, Npix, Nb, Nh)) return e;
    if (t->count != 6 || i < 0) return QFA_E_SIZE;
    const size_t want[6] = {(size_t)Npix * Nh, (size_t)Npix, (size_t)Nb, 1, 1, 1};
    AdamMultiArgs a;
    a.t = *t;
    unsigned nblk = 0;
    for (int k = 0; k < 6; ++k) {
        if (t->n[k] != want[k]) return QFA_E_SIZE;
        if (t->n[k] != 0 && (!t->p[k] || !t->m[k] || !t->v[k] || !t->p_out[k])) return QFA_E_NULL;
        a.blk0[k] = nblk;
        nblk += (unsigned)((t->n[k] + 255) / 256);
    }
    for (int k = 6; k <= QFA_ADAM_MAX; ++k) a.blk0[k] = nblk;
    const float bc1 = (float)(1.0 - pow(b1, (double)(i + 1))), bc2 = (float)(1.0 - pow(b2, (double)(i + 1)));
    k_finalize_adam<<<nblk, 256, 0, (hipStream_t)stream>>>(a, accum, Npix, Nb, Nh, loss, (float)lr, (float)b1, (float)b2,
                                                           (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, (float)wd, bc1, bc2);
    return hip_status();
}

int qfa_clip_f32(const float *x, float *y, size_t n, float lo, float hi, void *stream) {
    if (n == 0) return 0;                      // empty tensors (N_b = 0) carry NULL data pointers
    if (!x || !y) return QFA_E_NULL;
    k_clip<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, n, lo, hi);
    return hip_status();
}

int qfa_smooth_f32(const float *x, float *y, int n, int cols, int half, void *stream) {
    if (n == 0) return 0;                      // empty tensors (N_b = 0) carry NULL data pointers
    if (!x || !y) return QFA_E_NULL;
    if (n < 1 || cols < 1 || half < 0) return QFA_E_SIZE;
    const size_t tot = (size_t)n * cols;
    k_smooth<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, n, cols, half);
    return hip_status();
}

int qfa_tau_f32(const float *z, float *out, size_t n, const qfa_tau_t *tau, void *stream) {
    if (!z || !out || !tau) return QFA_E_NULL;
    if (n == 0) return 0;
    k_tau<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(z, out, n, *tau);
    return hip_status();
}

int qfa_tauhi_f32(const float *z, const float *tau0, const float *beta, float *out, size_t n, void *stream) {
    if (!z || !out || !tau0 || !beta) return QFA_E_NULL;
    if (n == 0) return 0;
    k_tauhi<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(z, tau0, beta, out, n);
    return hip_status();
}

int qfa_omega_func_f32(const float *z, const float *tau0, const float *beta, const float *c0, float *out, size_t n,
                       void *stream) {
    if (!z || !out || !tau0 || !beta || !c0) return QFA_E_NULL;
    if (n == 0) return 0;
    k_omega_func<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(z, tau0, beta, c0, out, n);
    return hip_status();
}

static int fill_lyman(int which, double wav0, LymanTable *t) {
    double amp, scale, expo, off;
    switch (which) {
        case QFA_TAU_BECKER: amp = 0.751; scale = 1.0 / 4.5; expo = 2.90; off = -0.132; break;
        case QFA_TAU_FG: amp = 0.0018; scale = 1.0; expo = 3.92; off = 0.0; break;
        case QFA_TAU_KAMBLE: amp = 5.54e-3; scale = 1.0; expo = 3.182; off = 0.0; break;
        case QFA_TAU_MOCK: amp = 0.2231435513142097; scale = 1.0 / 3.25; expo = 3.2; off = 0.0; break;
        default: return QFA_E_TAU;
    }
    t->amp = amp; t->scale = scale; t->expo = expo; t->offset = off;
    t->level = 0;
    for (int i = 0; i < 30; ++i) {
        t->lam[i] = kLymanLam[i];
        t->coeff[i] = kLymanLam[i] * kLymanF[i] / (kLymanLam[0] * kLymanF[0]);
        if (wav0 < kLymanLam[i]) t->level = i + 1;        // lambdas decrease (QFA/utils.py:186-192)
    }
    return t->level == 0 ? QFA_E_SIZE : 0;
}

int qfa_build_batch_f32(const float *flux, const float *error, const double *zqso, const int *idx, const double *wav,
                        double wav0, const double *mu, int which, int nrow, int Npix, int Nb, int64_t row_stride,
                        float *delta, float *error_out, float *zabs, uint8_t *mask, void *stream) {
    if (!flux || !error || !zqso || !wav || !mu || !delta || !error_out || !mask || (Nb > 0 && !zabs)) return QFA_E_NULL;
    if (nrow < 1 || Npix < 1 || Nb < 0 || Nb > Npix || (row_stride != 0 && row_stride < Npix)) return QFA_E_SIZE;
    LymanTable tab;
    if (int e = fill_lyman(which, wav0, &tab)) return e;
    const size_t tot = (size_t)nrow * Npix;
    k_build_batch<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        flux, error, zqso, idx, wav, mu, tab, nrow, Npix, Nb, (size_t)(row_stride ? row_stride : Npix), delta, error_out, zabs, mask);
    return hip_status();
}

int qfa_build_resident_f32(const float *flux, const float *error, const double *zqso, const double *wav, double wav0,
                           const double *mu, int which, int64_t nrow, int Npix, int Nb, int64_t row_stride, float *delta,
                           uint8_t *mask, float *zq1, void *stream) {
    if (!flux || !error || !zqso || !wav || !mu || !delta || !mask || !zq1) return QFA_E_NULL;
    if (nrow < 1 || Npix < 1 || Nb < 0 || Nb > Npix || row_stride < Npix || row_stride >= (1LL << 31)) return QFA_E_SIZE;
    LymanTable tab;
    if (int e = fill_lyman(which, wav0, &tab)) return e;
    const size_t tot = (size_t)nrow * (size_t)row_stride;
    if ((tot + 255) / 256 >= (1ull << 31)) return QFA_E_SIZE;
    k_build_resident<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(flux, error, zqso, wav, mu, tab, (size_t)nrow,
                                                                                Npix, Nb, (size_t)row_stride, delta, mask, zq1);
    return hip_status();
}

int qfa_zabs_factor_f32(const float *zabs, int B, int Nb, float tol, float *zq1, float *pix_ratio, unsigned *nbad,
                        void *stream) {
    if (!zabs || !zq1 || !pix_ratio || !nbad) return QFA_E_NULL;
    if (B < 1 || Nb < 1 || !(tol >= 0.f)) return QFA_E_SIZE;
    hipStream_t st = (hipStream_t)stream;
    const int n = B > Nb ? B : Nb;
    k_zfactor_derive<<<(n + 255) / 256, 256, 0, st>>>(zabs, B, Nb, zq1, pix_ratio, nbad);
    const dim3 grid((unsigned)((Nb + 255) / 256 < 8 ? (Nb + 255) / 256 : 8), (unsigned)((B + 7) / 8));
    k_zfactor_check<<<grid, 256, 0, st>>>(zabs, B, Nb, zq1, pix_ratio, tol, nbad);
    return hip_status();
}

int qfa_mu_estimate_f64(const float *flux, const float *error, const double *zqso, const double *wav, double wav0,
                        int which, int B, int Npix, int Nb, int64_t row_stride, int window_len, double *scratch,
                        double *mu_raw, double *mu_smooth, void *stream) {
    if (!flux || !error || !zqso || !wav || !scratch || !mu_raw) return QFA_E_NULL;
    if (B < 1 || Npix < 1 || Nb < 0 || Nb > Npix || window_len < 2 || window_len > Npix) return QFA_E_SIZE;
    if (row_stride != 0 && row_stride < Npix) return QFA_E_SIZE;
    LymanTable tab;
    if (int e = fill_lyman(which, wav0, &tab)) return e;
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(scratch, 0, 2 * (size_t)Npix * sizeof(double), st);
    const int chunk = 64;
    const dim3 grid((Npix + 255) / 256, (B + chunk - 1) / chunk);
    k_mu_accumulate<<<grid, 256, 0, st>>>(flux, error, zqso, wav, tab, B, Npix, Nb, (size_t)(row_stride ? row_stride : Npix), chunk,
                                          scratch, scratch + Npix);
    k_mu_finish<<<(Npix + 255) / 256, 256, 0, st>>>(scratch, scratch + Npix, Npix, window_len, mu_raw, mu_smooth);
    return hip_status();
}

int qfa_mu_sums_f64(const float *flux, const float *error, const double *zqso, const double *wav, double wav0, int which,
                    int B, int Npix, int Nb, int64_t row_stride, double *scratch, void *stream) {
    if (!flux || !error || !zqso || !wav || !scratch) return QFA_E_NULL;
    if (B < 1 || Npix < 1 || Nb < 0 || Nb > Npix || (row_stride != 0 && row_stride < Npix)) return QFA_E_SIZE;
    LymanTable tab;
    if (int e = fill_lyman(which, wav0, &tab)) return e;
    const int chunk = 64;
    const dim3 grid((Npix + 255) / 256, (B + chunk - 1) / chunk);
    k_mu_accumulate<<<grid, 256, 0, (hipStream_t)stream>>>(flux, error, zqso, wav, tab, B, Npix, Nb,
                                                           (size_t)(row_stride ? row_stride : Npix), chunk, scratch, scratch + Npix);
    return hip_status();
}

int qfa_mu_finish_f64(const double *scratch, int Npix, int window_len, double *mu_raw, double *mu_smooth, void *stream) {
    if (!scratch || !mu_raw) return QFA_E_NULL;
    if (Npix < 1 || window_len < 2 || window_len > Npix) return QFA_E_SIZE;
    k_mu_finish<<<(Npix + 255) / 256, 256, 0, (hipStream_t)stream>>>(scratch, scratch + Npix, Npix, window_len, mu_raw,
                                                                  mu_smooth);
    return hip_status();
}

int qfa_woodbury_f32(const float *M, const float *D, int n, int k, float *inv, float *logdet, void *workspace,
                     size_t workspace_bytes, void *stream) {
    if (!M || !D || !workspace || (!inv && !logdet)) return QFA_E_NULL;
    if (n < 1 || k < 1 || k > 32) return QFA_E_SIZE;
    if (workspace_bytes < (size_t)(k * k + 1) * sizeof(double)) return QFA_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double *core = (double *)workspace;
    k_wood_core<<<1, 1024, (size_t)k * k * sizeof(double), st>>>(M, D, n, k, core);
    const size_t tot = inv ? (size_t)n * n : 1;
    k_wood_inv<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(M, D, n, k, core, inv, logdet);
    return hip_status();
}

}  // extern "C"
