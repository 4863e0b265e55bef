/* qfa_hip.h -- C-ABI of libqfa_hip.so, the MI355X (gfx950) implementation of the QFA hot path.
 *
 * The reference (ZechangSun/QFA) is pure Python and has no FFI: the boundary it offers is the
 * Python method surface of QFA/model.py and QFA/optimizer.py.  qfa_amd keeps that surface in
 * Python and calls the functions below through ctypes (qfa_amd/_lib.py).  Every entry point
 * names the reference interface it replaces (file:line in the reference tree).
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (torch tensors: tensor.data_ptr());
 *     the library allocates nothing, reads no environment variable, keeps no mutable state (but the
 *     per-device compute-unit count, queried once) and is re-entrant across streams and devices;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - arrays are row-major, contiguous, float32; masks are 1 byte per pixel (torch.bool);
 *   - return value: 0 = ok, negative = invalid argument (QFA_E_*), positive = hipError_t;
 *     the library never throws and never calls exit();
 *   - B spectra, Npix = Nb + Nr pixels (blue side first), Nh latent factors (1..32).
 */
#ifndef QFA_HIP_H
#define QFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QFA_ABI_VERSION 4   /* v2: qfa_batch_t carries the factored-z input form; *_ex_f32 entry points with `flags`;
                             * v3: qfa_batch_t carries the resident, indexed input form (rows, row_stride);
                             * v4: qfa_zabs_factor_f32, QFA_E_FLAGS                                                  */

#define QFA_E_NULL      (-1)   /* a required pointer is NULL            */
#define QFA_E_SIZE      (-2)   /* B/Npix/Nb/Nh out of range              */
#define QFA_E_WORKSPACE (-3)   /* workspace smaller than qfa_workspace_bytes */
#define QFA_E_TAU       (-4)   /* unknown tau model                      */
#define QFA_E_FLAGS     (-5)   /* a `flags` bit that does not apply to this shape (QFA_F_S3_FAST at N_h <= 16) */

/* Mean-optical-depth model tau(z) = (amp * ((1+z)*scale)^expo + offset) * series_coeff
 * (reference QFA/utils.py:95-141, 149-171).  qfa_tau_model() fills it for the four built-ins. */
typedef struct {
    float amp;      /* already multiplied by the Lyman-series coefficient */
    float scale;
    float expo;
    float offset;   /* already multiplied by the Lyman-series coefficient */
} qfa_tau_t;

enum { QFA_TAU_BECKER = 0, QFA_TAU_FG = 1, QFA_TAU_KAMBLE = 2, QFA_TAU_MOCK = 3 };

/* Model parameters (reference QFA/model.py:37-55, property `parameters` :297-306). */
typedef struct {
    const float *F;      /* (Npix, Nh) */
    const float *Psi;    /* (Npix,)    */
    const float *omega;  /* (Nb,)      */
    const float *tau0;   /* scalar on device */
    const float *c0;     /* scalar on device */
    const float *beta;   /* scalar on device */
} qfa_params_t;

/* Spectra batch (contract of Dataloader.next_batch, reference QFA/dataloader.py:124-138). */
typedef struct {
    const float   *delta;   /* (B, Npix)  delta = flux - mu*A for training; raw flux for predict */
    const float   *error;   /* (B, Npix)  */
    const float   *zabs;    /* (B, Nb)    */
    const uint8_t *mask;    /* (B, Npix)  1 = pixel is used; masked pixels may hold -999 */
    const float   *A_blue;  /* optional (B, Nb): host-supplied exp(-tau(zabs)) for a custom tau
                               callable (reference QFA/model.py:26,43); NULL = use `tau` */
    /* Factored-z input form (ABI v2).  The reference's loader builds zabs[s][i] = (1 + z_qso[s]) wav[i] / 1215.67 - 1
     * (QFA/dataloader.py:102): 1 + zabs is the product of a per-spectrum and a per-pixel factor.  A caller that has
     * them passes zq1 = 1 + z_qso (B,) and pix_ratio = wav_blue / 1215.67 (Nb,) and may leave zabs NULL: the (B, Nb)
     * array is then never read (4 Nb bytes per spectrum and pass) and (1+z)^beta, tau(z) become products of a
     * per-spectrum and a per-pixel term (three transcendentals per blue element instead of six).  Both NULL = read
     * zabs.  Not combined with A_blue (a custom tau callable is evaluated on zabs by the caller). */
    const float   *zq1;        /* optional (B,)  */
    const float   *pix_ratio;  /* optional (Nb,) */
    /* Resident, indexed input form (ABI v3).  The reference materialises every batch on the host
     * (QFA/dataloader.py:124-138) and shuffles by permuting the whole data set (:154-167).  Here the data set can stay
     * where it is in HBM: spectrum s of the batch is row  r = rows ? rows[s] : s  of delta / error / mask -- rows
     * `row_stride` ELEMENTS apart (4 row_stride bytes for delta and error, row_stride bytes for the mask) -- and of zabs
     * (rows Nb apart) and zq1, so that a batch is B indices into arrays that are built once and a shuffled epoch is one
     * permutation on the device.  row_stride = 0 means Npix (contiguous rows); a loader that pads its rows to a multiple of
     * 32 elements (128 bytes) gives every row the alignment the kernels' 128-byte row segments like (N_pix = 1913, 9243:
     * 4 - 12 % of passes 1 and 2).  Pixels [Npix, row_stride) of a row are never read.  Outputs (nll, ll, hmean, hcov,
     * cont, unc) are in batch order and contiguous.  Not combined with A_blue. */
    const int32_t *rows;       /* optional (B,) device array of row indices, each >= 0 */
    int64_t        row_stride; /* 0 = Npix; else >= Npix */
} qfa_batch_t;

/* `flags` of the *_ex_f32 entry points (0 = the defaults; A/B timing and the cross-checks in tests/). */
#define QFA_F_PASS2_F32    0x1u  /* N_h <= 16: pass 2 in its float32-MFMA form (k_grads; never the default there)    */
#define QFA_F_PASS2_XDL    0x2u  /* N_h <= 16: pass 2 in its two-role all-XDL form (k_grads_x: 64 spectra per workgroup
                                  * walk the pixel axis)                                                         */
#define QFA_F_S3_FAST      0x4u  /* N_h = 17..32: stage 3 of pass 2 (k_grads_s3) with three bf16 piece products (operands
                                    carried to ~17 bits, <= 1.1e-5 per product).  Kept for callers that set it; since
                                    round 5 the default form runs three FLOAT16 piece products at float32 grade and is
                                    as fast.  At N_h <= 16 there is no such form: the call returns QFA_E_FLAGS        */
#define QFA_F_PREDICT_F32  0x8u  /* posterior writer in its float32-MFMA form (k_predict_out)                   */
#define QFA_F_SYNC         0x20u /* debugging: synchronise `stream` before returning, so that an asynchronous fault of
                                    THIS call's kernels is returned by THIS call (positive hipError_t) instead of
                                    surfacing at the caller's next synchronisation without context               */
#define QFA_F_PASS2_PIXRES 0x40u /* N_h <= 16: pass 2 in its pixel-resident all-XDL form (k_grads_t: a wave owns 16 pixels
                                  * and walks the spectra; the per-spectrum operands stream through LDS).  Without any
                                  * QFA_F_PASS2_* flag the library picks between k_grads_x (small batches) and this form
                                  * (from 96 spectra per CU on; N_h <= 8 and N_pix >= 1024: from 36 per CU on) --
                                  * qfa_host.h, pass2_use_pixres                                                      */
#define QFA_F_ZERO_ACCUM   0x80u /* qfa_nll_grad_*: the library zeroes `accum` itself before it adds to it (inside the first
                                  * kernel of the call: one launch less than a caller-side fill -- the step of a small
                                  * batch is a chain of dependent launches a few microseconds long)                  */
/* (0x10: the one-wave-per-SIMD form k_grads_w of round 3, removed from the library in round 4 -- same results, 5.4 against
 *  2.15 ms at c3; the measurement is kept in profiles/r3_ablation_pass2.txt) */

int qfa_abi_version(void);

/* fills `out` for which in QFA_TAU_*, series 1..30 (reference QFA/utils.py:149-171,
 * coefficients QFA/Lyman_series.csv:2-31). */
int qfa_tau_model(int which, int series, qfa_tau_t *out);

/* bytes of scratch needed by qfa_nll_grad_f32 / qfa_predict_f32 for this shape */
size_t qfa_workspace_bytes(int B, int Npix, int Nh);

/* number of floats in the packed accumulation buffer used by qfa_nll_grad_f32 and
 * qfa_finalize_grads_f32:  [accF Npix*Nh | sumA Npix | gPsi Npix | gOmega Nb | cnt Npix | 8 scalars]
 * scalars = {g_tau0, g_c0, g_beta, n_spectra_with_blue, sum_nll, n_spectra, 0, 0}.
 * This is the buffer a data-parallel job all-reduces (sum) across ranks once per step. */
size_t qfa_accum_floats(int Npix, int Nb, int Nh);

/* Replaces the loop body of QFA.forward (reference QFA/model.py:98-103) and
 * QFA.loglikelihood_and_gradient_for_single_spectra (:107-158) together with MatrixInverse /
 * MatrixLogDet (QFA/utils.py:12-54) for a whole batch.
 *   nll   (B,)  per-spectrum negative log-likelihood (out, may be NULL)
 *   accum qfa_accum_floats() floats, ADDED to (caller zeroes it once per step).
 * The per-spectrum sums are raw (not normalised): see qfa_finalize_grads_f32.
 * Limit: the counts in `accum` (cnt per pixel, n_spectra, n_spectra_with_blue) are float32 sums of ones, exact up to
 * 2^24: B > 16 777 216 returns QFA_E_SIZE, and a caller that adds several launches (or ranks) into one buffer must
 * finalise before the total number of spectra passes 2^24. */
int qfa_nll_grad_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau,
                     int B, int Npix, int Nb, int Nh,
                     float *nll, float *accum, void *workspace, size_t workspace_bytes,
                     void *stream);

/* Same call with stage timing for benchmarks: `events` is NULL or an array of 5 hipEvent_t
 * (entries may be NULL) recorded on `stream` at {start, after PF-image build, after pass 1
 * (moments), after the k x k solve, after pass 2 (gradients)}. */
int qfa_nll_grad_events_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau,
                            int B, int Npix, int Nb, int Nh,
                            float *nll, float *accum, void *workspace, size_t workspace_bytes,
                            void *stream, void *const *events);

/* Deterministic accumulation (SURVEY.md section 7 hard part 5, 8(e)).  By default the tiles of a block of 64 spectra
 * are added to `accum` with float32 atomics, whose arrival order -- and so the last bits of the sums -- changes
 * from run to run.  With a slab of qfa_det_slab_bytes() bytes every block writes its tile partials to its own
 * row of the slab (plain stores) and a reducer adds the rows to `accum` in block order: two runs on the same
 * inputs (and the same B, which fixes the work plan) are bit-identical.  slab == NULL is qfa_nll_grad_events_f32.
 * (Where pass 2 runs in its pixel-resident form -- from 96 spectra per CU on, 36 at N_h <= 8, or QFA_F_PASS2_PIXRES --
 * the per-range sums always leave through such rows, inside the workspace when slab == NULL: no float atomics
 * there in either mode.) */
size_t qfa_det_slab_bytes(int B, int Npix, int Nb, int Nh);
int qfa_nll_grad_det_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau,
                         int B, int Npix, int Nb, int Nh,
                         float *nll, float *accum, void *workspace, size_t workspace_bytes,
                         void *slab, size_t slab_bytes, void *stream, void *const *events);

/* The general form of the three calls above: slab may be NULL (then slab_bytes is ignored), events may be NULL,
 * flags = QFA_F_* (0 = defaults). */
int qfa_nll_grad_ex_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau,
                        int B, int Npix, int Nb, int Nh,
                        float *nll, float *accum, void *workspace, size_t workspace_bytes,
                        void *slab, size_t slab_bytes, unsigned flags, void *stream, void *const *events);

/* Replaces the normalisation of QFA.forward (reference QFA/model.py:104): elementwise
 * grad = sum / count (0/0 = NaN), loss = sum_nll / n_spectra (model.py:100).  Reads `accum`
 * (after the optional all-reduce) and writes gradients with the reference's shapes.
 * normalize = 0 returns the raw sums instead (the per-spectrum gradient of
 * loglikelihood_and_gradient_for_single_spectra when accum holds one spectrum). */
int qfa_finalize_grads_f32(const float *accum, const float *F, int Npix, int Nb, int Nh,
                           int normalize, float *gF, float *gPsi, float *gOmega, float *gTau0,
                           float *gC0, float *gBeta, float *loss, void *stream);

/* Replaces QFA.prediction_for_single_spectra (reference QFA/model.py:160-180) for a batch:
 * b->delta holds the raw flux.  Outputs: ll (B,), hmean (B,Nh), hcov (B,Nh,Nh), cont (B,Npix),
 * unc (B,Npix): contiguous, any 4-byte alignment.  (The writer is fastest when the rows of cont / unc start on
 * 64-byte boundaries; at N_h <= 8 it stores whole aligned lines for any N_pix as long as cont and unc are
 * congruent modulo 128 bytes, as two allocations are.) */
int qfa_predict_f32(const qfa_params_t *p, const float *mu, const qfa_batch_t *b,
                    const qfa_tau_t *tau, int B, int Npix, int Nb, int Nh,
                    float *ll, float *hmean, float *hcov, float *cont, float *unc,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Same call with stage timing for benchmarks: `events` is NULL or an array of 4 hipEvent_t (entries
 * may be NULL) recorded on `stream` at {start, after the parameter images + pass 1 (moments),
 * after the k x k solve (ll, hmean, hcov written), after the continuum / uncertainty writer}. */
int qfa_predict_events_f32(const qfa_params_t *p, const float *mu, const qfa_batch_t *b,
                           const qfa_tau_t *tau, int B, int Npix, int Nb, int Nh,
                           float *ll, float *hmean, float *hcov, float *cont, float *unc,
                           void *workspace, size_t workspace_bytes, void *stream, void *const *events);

int qfa_predict_ex_f32(const qfa_params_t *p, const float *mu, const qfa_batch_t *b,
                       const qfa_tau_t *tau, int B, int Npix, int Nb, int Nh,
                       float *ll, float *hmean, float *hcov, float *cont, float *unc,
                       void *workspace, size_t workspace_bytes, unsigned flags, void *stream, void *const *events);

/* Replaces Adam.update (reference QFA/optimizer.py:37-52) followed by the clamp of QFA.clip
 * (QFA/model.py:233-241) for ONE tensor of n elements:
 *   g' = g + wd*p; m = (1-b1) g' + b1 m; v = (1-b2) g'^2 + b2 v;
 *   p_out = clamp(p - lr * (m/bc1) / (sqrt(v/bc2) + eps), lo, hi),  bc = 1 - b^(i+1),
 * with i = Adam.i (advanced by Adam.step(), once per epoch in QFA.train).  The hyper-parameters
 * are doubles because the reference holds them as Python floats and rounds (1-b), b^(i+1) etc.
 * to float32 only when they meet a tensor.  m and v are updated in place; pass lo > hi to skip
 * the clamp.  NaN propagates as in torch. */
int qfa_adam_clip_f32(const float *p, const float *g, float *m, float *v, float *p_out, size_t n,
                      double lr, double b1, double b2, double eps, double wd, int i,
                      float lo, float hi, void *stream);

/* Adam.update + clip for up to QFA_ADAM_MAX tensors in ONE launch (reference QFA/optimizer.py:37-52 loops over the
 * parameter dict, QFA/model.py:233-241 clips each entry): same arithmetic as qfa_adam_clip_f32 per tensor.
 * lo[k] > hi[k] disables the clip of tensor k; p_out[k] may equal p[k] (in place). */
#define QFA_ADAM_MAX 8
typedef struct {
    const float *p[QFA_ADAM_MAX];
    const float *g[QFA_ADAM_MAX];
    float *m[QFA_ADAM_MAX];
    float *v[QFA_ADAM_MAX];
    float *p_out[QFA_ADAM_MAX];
    size_t n[QFA_ADAM_MAX];
    float lo[QFA_ADAM_MAX], hi[QFA_ADAM_MAX];
    int count;
} qfa_adam_multi_t;
int qfa_adam_clip_multi_f32(const qfa_adam_multi_t *t, double lr, double b1, double b2, double eps, double wd, int i,
                            void *stream);

/* qfa_finalize_grads_f32 (normalised) followed by qfa_adam_clip_multi_f32 for the six parameter tensors in ONE launch -- the
 * reference's `loss, grad = self.forward(...); self.parameters = optimizer.update(self.parameters, grad)` (QFA/model.py:212-214)
 * without the gradients in between.  `t` holds the tensors in the order F, Psi, omega, tau0, c0, beta (count = 6; its `g`
 * pointers are not read); the gradient of every element is formed from `accum` with k_finalize's arithmetic and fed to
 * qfa_adam_clip_multi_f32's: the new parameters are bit-identical to the two calls.  loss (1 float) = sum NLL / n_spectra. */
int qfa_finalize_adam_clip_f32(const float *accum, int Npix, int Nb, int Nh, const qfa_adam_multi_t *t, double lr, double b1,
                               double b2, double eps, double wd, int i, float *loss, void *stream);

/* Replaces QFA.clip for one tensor (reference QFA/model.py:233-241): y = clamp(x, lo, hi), NaN kept. */
int qfa_clip_f32(const float *x, float *y, size_t n, float lo, float hi, void *stream);

/* Replaces QFA.smooth (reference QFA/model.py:243-252): edge-aware moving average of
 * (2*half+1) rows along axis 0 of an (n, cols) array, divisor = in-range sample count. */
int qfa_smooth_f32(const float *x, float *y, int n, int cols, int half, void *stream);

/* Replace tau(), tauHI(), omega_func() (reference QFA/utils.py:57-92, 149-171), elementwise on n. */
int qfa_tau_f32(const float *z, float *out, size_t n, const qfa_tau_t *tau, void *stream);
int qfa_tauhi_f32(const float *z, const float *tau0, const float *beta, float *out, size_t n,
                  void *stream);
int qfa_omega_func_f32(const float *z, const float *tau0, const float *beta, const float *c0,
                       float *out, size_t n, void *stream);

/* Device-side batch builder.  Replaces what Dataloader.next_batch / __init__ do
 * on the host with numpy (reference QFA/dataloader.py:29,102,124-138 and tau_total, QFA/utils.py:174-203):
 * for row r of the batch, spectrum s = idx ? idx[r] : r of the resident flux/error arrays (N rows, row_stride elements
 * apart; 0 = Npix),
 *   zabs  = (1+zqso) wav_blue / 1215.67 - 1,  delta = flux - mu * exp(-tau_total) (blue) | flux - mu (red),
 *   mask  = (flux != -999) & (error != -999),  error_out = error[s].
 * float64 arithmetic like numpy, outputs (contiguous, batch order) rounded to float32 once.  wav0 = wav[0] (host copy). */
int qfa_build_batch_f32(const float *flux, const float *error, const double *zqso, const int *idx,
                        const double *wav, double wav0, const double *mu, int which, int nrow, int Npix,
                        int Nb, int64_t row_stride, float *delta, float *error_out, float *zabs, uint8_t *mask,
                        void *stream);

/* The resident form of a whole data set, built ONCE per loader instead of once per batch (ABI v3; the reference
 * recomputes delta for every batch, QFA/dataloader.py:135-136, although it depends on mu and tau only): for every
 * row r < nrow of flux / error (rows row_stride >= Npix elements apart)
 *   delta[r] = flux - mu * exp(-tau_total) (blue) | flux - mu (red),  mask[r] = (flux != -999) & (error != -999),
 * written with the SAME row stride (pad pixels: 0 / masked), and zq1[r] = (float)(1 + zqso[r]).  The same arithmetic as
 * qfa_build_batch_f32: a batch of the resident form -- qfa_batch_t{delta, error, mask, zq1, pix_ratio, rows, row_stride} --
 * holds bit for bit the numbers the materialised batch of the same rows holds. */
int qfa_build_resident_f32(const float *flux, const float *error, const double *zqso, const double *wav, double wav0,
                           const double *mu, int which, int64_t nrow, int Npix, int Nb, int64_t row_stride,
                           float *delta, uint8_t *mask, float *zq1, void *stream);

/* ABI v4.  Does a caller's zabs (B, Nb) have the structure the reference's loader gives it -- 1 + zabs[s][i] =
 * (1 + z_qso[s]) wav_i / 1215.67 (reference QFA/dataloader.py:102) -- so that the factored-z input form (qfa_batch_t::zq1 /
 * pix_ratio above) may stand in for it?  Writes zq1[s] = 1 + zabs[s][0] (B floats) and pix_ratio[i] = (1 + zabs[0][i]) /
 * (1 + zabs[0][0]) (Nb floats; the quotient in float64, rounded once) and counts in *nbad (device memory, zeroed by the call)
 * the elements with |(1 + zabs[s][i]) - zq1[s] pix_ratio[i]| > tol (1 + zabs[s][i]) (a NaN counts).  nbad == 0: every 1 + z the
 * factored kernels form is within tol (relative) of the one the zabs kernels read -- at tol = 4e-7 (three float32 roundings) the
 * results agree as the two forms of one loader's batch do (tests/test_hip_parity.py).  One pass over zabs: B Nb 4 bytes read.
 * Asynchronous like every entry point: the caller reads *nbad behind the stream. */
int qfa_zabs_factor_f32(const float *zabs, int B, int Nb, float tol, float *zq1, float *pix_ratio, unsigned *nbad,
                        void *stream);

/* Replaces the continuum-mean estimate of Dataloader.__init__ (reference QFA/dataloader.py:110-112):
 * mu_raw = sum_s flux exp(+tau_total) mask / #(flux != -999); mu_smooth = reflect-padded boxcar of
 * window_len (QFA/utils.py:206-219; may be NULL).  scratch: 2*Npix doubles.  flux / error rows row_stride elements apart
 * (0 = Npix). */
int qfa_mu_estimate_f64(const float *flux, const float *error, const double *zqso, const double *wav,
                        double wav0, int which, int B, int Npix, int Nb, int64_t row_stride, int window_len,
                        double *scratch, double *mu_raw, double *mu_smooth, void *stream);

/* The two halves of qfa_mu_estimate_f64 for a data-parallel loader (each rank holds a shard of the
 * spectra): qfa_mu_sums_f64 ADDS this shard's per-pixel sums to scratch = [num Npix | den Npix]
 * (caller zeroes it), the caller all-reduces scratch over the ranks, qfa_mu_finish_f64 divides and
 * smooths.  Same arithmetic as the one-call form. */
int qfa_mu_sums_f64(const float *flux, const float *error, const double *zqso, const double *wav,
                    double wav0, int which, int B, int Npix, int Nb, int64_t row_stride, double *scratch,
                    void *stream);
int qfa_mu_finish_f64(const double *scratch, int Npix, int window_len, double *mu_raw, double *mu_smooth,
                      void *stream);

/* Replace MatrixInverse / MatrixLogDet (reference QFA/utils.py:12-54) for one (n,k) M and (n,) D:
 * inv (n,n) dense, logdet scalar (Cholesky-free Gauss-Jordan on the k x k core, finite where the
 * reference's float32 det overflows). workspace: qfa_workspace_bytes(1, n, k). */
int qfa_woodbury_f32(const float *M, const float *D, int n, int k, float *inv, float *logdet,
                     void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QFA_HIP_H */
