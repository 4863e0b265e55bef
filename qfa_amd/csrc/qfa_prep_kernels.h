// qfa_prep_kernels.h -- device-side batch builder: what the reference's
// Dataloader does on the host with numpy for every batch (QFA/dataloader.py:102,110-112,124-138;
// QFA/utils.py:174-219), here on the spectra resident in HBM.  Arithmetic in float64 like numpy;
// outputs are rounded to float32 once, where the reference builds its float32 tensors.
#pragma once
#include "qfa_common.h"

struct LymanTable {
    double lam[30];
    double coeff[30];
    double amp, scale, expo, offset;      // tau(z) = (amp ((1+z) scale)^expo + offset) * coeff
    int level;                            // number of series lines redward of wav[0]
};

__device__ __forceinline__ double tau_total_px(const LymanTable &t, double opz_wav /* (1+zqso)*wav */, double wav) {
    double tot = 0.0;
    for (int i = 0; i < t.level; ++i) {
        if (wav < t.lam[i]) {
            const double opz = opz_wav / t.lam[i];            // 1 + z_abs of series line i
            tot += (t.amp * pow(opz * t.scale, t.expo) + t.offset) * t.coeff[i];
        }
    }
    return tot;
}

// One thread per (row of the batch, pixel).  idx (may be NULL) selects the spectrum of each row.
__global__ void k_build_batch(const float *__restrict__ flux, const float *__restrict__ error,
                              const double *__restrict__ zqso, const int *__restrict__ idx,
                              const double *__restrict__ wav, const double *__restrict__ mu, LymanTable tab, int nrow,
                              int Npix, int Nb, size_t in_stride, float *__restrict__ delta, float *__restrict__ err_out,
                              float *__restrict__ zabs, uint8_t *__restrict__ mask) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)nrow * Npix) return;
    const int r = (int)(g / Npix), i = (int)(g % Npix);
    const int s = idx ? idx[r] : r;
    const float f = flux[(size_t)s * in_stride + i], e = error[(size_t)s * in_stride + i];
    const double w = wav[i];
    const double opzw = (zqso[s] + 1.0) * w;
    double trans = 1.0;
    if (i < Nb) {
        trans = exp(-tau_total_px(tab, opzw, w));
        zabs[(size_t)r * Nb + i] = (float)(opzw / 1215.67 - 1.0);          // dataloader.py:102
    }
    delta[g] = (float)((double)f - mu[i] * trans);                          // dataloader.py:135-136
    err_out[g] = e;
    mask[g] = (f != -999.f) && (e != -999.f);                               // dataloader.py:29
}

// The resident form of a data set (ABI v3; qfa_build_resident_f32): every row of flux / error (rows `stride` elements apart)
// -> delta and mask rows with the same stride (pad pixels zeroed / masked), zq1 = 1 + z_qso.  Same arithmetic as k_build_batch;
// zabs is not written (the kernels take the factored form zq1 x pix_ratio).  One thread per (row, pixel of the padded row).
__global__ void k_build_resident(const float *__restrict__ flux, const float *__restrict__ error,
                                 const double *__restrict__ zqso, const double *__restrict__ wav,
                                 const double *__restrict__ mu, LymanTable tab, size_t nrow, int Npix, int Nb, size_t stride,
                                 float *__restrict__ delta, uint8_t *__restrict__ mask, float *__restrict__ zq1) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nrow * stride) return;
    const size_t s = g / stride;
    const int i = (int)(g % stride);
    if (i >= Npix) {
        delta[g] = 0.f;
        mask[g] = 0;
        return;
    }
    const float f = flux[g], e = error[g];
    const double w = wav[i];
    const double opzw = (zqso[s] + 1.0) * w;
    double trans = 1.0;
    if (i < Nb) trans = exp(-tau_total_px(tab, opzw, w));
    delta[g] = (float)((double)f - mu[i] * trans);                          // dataloader.py:135-136
    mask[g] = (f != -999.f) && (e != -999.f);                               // dataloader.py:29
    if (i == 0) zq1[s] = (float)(zqso[s] + 1.0);
}

// mu estimate: per pixel, sums over spectra.  grid.x over pixels, grid.y over chunks of spectra.
__global__ void k_mu_accumulate(const float *__restrict__ flux, const float *__restrict__ error,
                                const double *__restrict__ zqso, const double *__restrict__ wav, LymanTable tab, int B,
                                int Npix, int Nb, size_t in_stride, int chunk, double *__restrict__ num, double *__restrict__ den) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Npix) return;
    const int s0 = blockIdx.y * chunk, s1 = min(B, s0 + chunk);
    const double w = wav[i];
    double a = 0.0, c = 0.0;
    for (int s = s0; s < s1; ++s) {
        const float f = flux[(size_t)s * in_stride + i], e = error[(size_t)s * in_stride + i];
        const bool m = (f != -999.f) && (e != -999.f);
        double up = 1.0;
        if (i < Nb) up = exp(tau_total_px(tab, (zqso[s] + 1.0) * w, w));
        if (m) a += (double)f * up;                                         // dataloader.py:110-111
        c += (f != -999.f) ? 1.0 : 0.0;
    }
    atomicAdd(num + i, a);
    atomicAdd(den + i, c);
}

// raw = num / den, then the reference's reflect-padded boxcar (QFA/utils.py:206-219)
__global__ void k_mu_finish(const double *__restrict__ num, const double *__restrict__ den, int n, int window,
                            double *__restrict__ raw, double *__restrict__ smooth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    raw[i] = num[i] / den[i];
    if (!smooth) return;
    // extended signal e = [s[w-1], ..., s[1], s[0..n-1], s[n-2], ..., s[n-w]], y = boxcar(e) 'valid',
    // output i = y[i + w/2 - 1] = mean(e[i + w/2 - 1 .. i + w/2 - 1 + w - 1])
    const int start = i + window / 2 - 1;
    double acc = 0.0;
    for (int k = 0; k < window; ++k) {
        int p = start + k - (window - 1);          // index into s, may be out of range -> reflect
        if (p < 0) p = -p;
        else if (p >= n) p = 2 * (n - 1) - p;
        acc += num[p] / den[p];
    }
    smooth[i] = acc / window;
}
