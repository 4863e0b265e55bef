// qfa_k32.hip -- the N_h = 17..32 instantiations of the step kernels (see qfa_host.h for why they live
// in their own translation unit).
#include "qfa_host.h"

int qfa_k32_nll_grad(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                     float *nll, float *accum, float *ws, hipStream_t st, void *const *events, void *slab, unsigned flags) {
    return run_nll_grad<32>(p, b, tau, B, Npix, Nb, Nh, nll, accum, ws, st, events, slab, flags);
}

int qfa_k32_predict(const qfa_params_t &p, const float *mu, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix,
                    int Nb, int Nh, float *ll, float *hmean, float *hcov, float *cont, float *unc, float *ws,
                    hipStream_t st, void *const *events, unsigned flags) {
    return run_predict<32>(p, mu, b, tau, B, Npix, Nb, Nh, ll, hmean, hcov, cont, unc, ws, st, events, flags);
}
