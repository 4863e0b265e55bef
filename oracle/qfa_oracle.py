"""CPU oracle for the QFA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path (``qfa_amd``) never
imports it and fails loudly when the HIP library is missing.

This is a from-scratch numpy restatement of the algorithm in the reference
(ZechangSun/QFA, MIT licence) in its O(N_pix * N_h^2) low-rank form:
masks are zero weights on full-length arrays, the N_pix x N_pix inverse is
never formed, and the only factorisation is the N_h x N_h Cholesky of
``C = I + M^T D^-1 M``.  Every function cites the reference lines it follows.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this module
against (a) the known-answer vectors the reference ships in
``data/spec-4321-55504-0114.npz`` (keys ll/h/our and ll_red/h_red/our_red) and
(b) fixtures produced by importing the reference itself in the build container
(``tests/golden/make_golden.py``).

Default arithmetic is float64 (the reference is float32; the float64 result is
the one both the reference and the HIP path are compared against, and it stays
finite for N_h >= 16 where the reference's float32 ``det`` overflows,
QFA/utils.py:54).
"""
from __future__ import annotations

import numpy as np

LOG2PI = 1.8378770664093453          # QFA/model.py:20
LYA = 1215.67                        # QFA/dataloader.py:15

# Lyman-series oscillator strengths and wavelengths (QFA/Lyman_series.csv:2-31).
# Only the ratio lambda*f / (lambda_alpha*f_alpha) is used (QFA/utils.py:146-147).
_LYMAN_F = np.array([
    4.1620e-01, 7.9140e-02, 2.9010e-02, 1.3950e-02, 7.8030e-03, 4.8160e-03,
    3.1850e-03, 2.2170e-03, 1.6060e-03, 1.2010e-03, 9.2190e-04, 7.2310e-04,
    5.7770e-04, 4.6890e-04, 3.8580e-04, 3.2120e-04, 2.7030e-04, 2.2970e-04,
    1.9680e-04, 1.6990e-04, 1.4770e-04, 1.2930e-04, 1.1370e-04, 1.0060e-04,
    8.9360e-05, 7.9780e-05, 7.1480e-05, 6.4350e-05, 5.8120e-05, 5.2640e-05])
_LYMAN_LAM = np.array([
    1215.6701, 1025.7222, 972.5367, 949.7430, 937.8034, 930.7482, 926.2256,
    923.1503, 920.9630, 919.3513, 918.1293, 917.1805, 916.4291, 915.8238,
    915.3289, 914.9192, 914.5762, 914.2861, 914.0385, 913.8256, 913.6411,
    913.4803, 913.3391, 913.2146, 913.1042, 913.0059, 912.9179, 912.8389,
    912.7676, 912.7032])
LYMAN_COEFF = _LYMAN_LAM * _LYMAN_F / (_LYMAN_LAM[0] * _LYMAN_F[0])

# tau(z) = amp * ((1+z) * scale) ** expo + offset, all times the series coeff.
# becker QFA/utils.py:105-106, fg :119-120, kamble :133-134, mock :141.
TAU_MODELS = {
    "becker": (0.751, 1.0 / 4.5, 2.90, -0.132),
    "fg": (0.0018, 1.0, 3.92, 0.0),
    "kamble": (5.54e-3, 1.0, 3.182, 0.0),
    "mock": (0.2231435513142097, 1.0 / 3.25, 3.2, 0.0),
}


def tau_eff(z, which="becker", series=1):
    """Mean Lyman optical depth (QFA/utils.py:149-171)."""
    if which not in TAU_MODELS:
        raise NotImplementedError(which)
    amp, scale, expo, offset = TAU_MODELS[which]
    z = np.asarray(z)
    return (amp * ((1.0 + z) * scale) ** expo + offset) * LYMAN_COEFF[series - 1]


def tau_hi(z, tau0, beta):
    """tau0 (1+z)^beta (QFA/utils.py:57-73)."""
    return tau0 * (1.0 + np.asarray(z)) ** beta


def omega_zdep(z, tau0, beta, c0):
    """(1 - c0 - exp(-tau0 (1+z)^beta))^2 (QFA/utils.py:76-92)."""
    r = 1.0 - c0 - np.exp(-tau_hi(z, tau0, beta))
    return r * r


def _as_params(params, dtype):
    out = {}
    for k in ("F", "Psi", "omega", "tau0", "c0", "beta"):
        out[k] = np.asarray(params[k], dtype=dtype)
    return out


def pixel_terms(params, error, zabs, tau_which="becker", tau_series=1, dtype=np.float64,
                A_blue=None):
    """Full-length per-pixel quantities of one spectrum.

    A (QFA/model.py:125), zdep (:129), D = A^2 Psi + omega*zdep + sigma^2 (:128-131).
    Returns A, zdep (zero on the red side), D, and 1+z on the blue side.
    """
    p = _as_params(params, dtype)
    Npix = p["F"].shape[0]
    Nb = p["omega"].shape[0]
    z = np.asarray(zabs, dtype=dtype)
    A = np.ones(Npix, dtype=dtype)
    if A_blue is None:
        A[:Nb] = np.exp(-tau_eff(z, tau_which, tau_series).astype(dtype))
    else:
        A[:Nb] = np.asarray(A_blue, dtype=dtype)
    zdep = np.zeros(Npix, dtype=dtype)
    zdep[:Nb] = omega_zdep(z, p["tau0"], p["beta"], p["c0"])
    om = np.zeros(Npix, dtype=dtype)
    om[:Nb] = p["omega"] * zdep[:Nb]
    sig = np.asarray(error, dtype=dtype)
    D = A * p["Psi"] * A + om + sig * sig
    return A, zdep, D


def _lowrank_core(F, A, D, w, delta):
    """Shared Woodbury pieces (QFA/utils.py:29-32, 51-54)."""
    k = F.shape[1]
    wD = np.where(w, 1.0 / np.where(w, D, 1.0), 0.0)
    d = np.where(w, delta, 0.0)
    M = A[:, None] * F
    C = np.eye(k, dtype=F.dtype) + (M * wD[:, None]).T @ M
    L = np.linalg.cholesky(C)
    b = M.T @ (wD * d)
    y = np.linalg.solve(C, b)
    u = wD * (d - M @ y)
    n = float(np.sum(w))
    logdet = float(np.sum(np.where(w, np.log(np.where(w, D, 1.0)), 0.0))
                   + 2.0 * np.sum(np.log(np.diag(L))))
    nll = 0.5 * (float(d @ u) + n * LOG2PI + logdet)
    return wD, d, M, C, y, u, nll


def nll_and_grads_single(params, delta, error, zabs, mask, tau_which="becker", tau_series=1,
                         dtype=np.float64, A_blue=None, return_abs=False):
    """One spectrum: negative log-likelihood and the reference's six 'gradients'.

    Follows QFA/model.py:107-158 in low-rank form.
    The F / tau0 / beta / c0 expressions are the reference's formulas, which are
    not the true derivatives (quirk Q2); rows of masked pixels are zero.
    """
    p = _as_params(params, dtype)
    F = p["F"]
    Nb = p["omega"].shape[0]
    w = np.asarray(mask, dtype=bool)
    A, zdep, D = pixel_terms(params, error, zabs, tau_which, tau_series, dtype, A_blue)
    wD, d, M, C, y, u, nll = _lowrank_core(F, A, D, w, np.asarray(delta, dtype=dtype))

    Cinv = np.linalg.inv(C)
    # diag(Sigma^-1)_i = w/D - (wA/D)^2 f^T C^-1 f     (model.py:132,138)
    q = np.einsum("ia,ab,ib->i", F, Cinv, F)
    dS = wD - (wD * A) ** 2 * q
    dG = 0.5 * (dS - u * u)                           # diag of partialSigma, model.py:136,138
    # partialF = 2 diag(A) G diag(A) M, G = (Sigma^-1 - u u^T)/2   (model.py:136-137)
    X = (A * A)[:, None] * F                          # diag(A) M
    T = (M * wD[:, None]).T @ X
    Z = Cinv @ T
    pvec = X.T @ u
    gF = A[:, None] * (wD[:, None] * X - wD[:, None] * (M @ Z) - u[:, None] * pvec[None, :])
    gF = np.where(w[:, None], gF, 0.0)
    gPsi = np.where(w, A * A * dG, 0.0)               # model.py:139
    gOm = np.where(w[:Nb], dG[:Nb] * zdep[:Nb], 0.0)  # model.py:140
    z = np.asarray(zabs, dtype=dtype)
    pw = (1.0 + z) ** p["beta"]
    root = 1.0 - p["tau0"] * pw - p["c0"]             # model.py:141 (no exp, unlike zdep)
    e = np.where(w[:Nb], dG[:Nb] * (p["omega"] * zdep[:Nb]) * zdep[:Nb] * 2.0 * root, 0.0)
    g_tau0 = -np.sum(e * pw)                          # model.py:142
    g_beta = -np.sum(e * p["tau0"] * pw * np.log(1.0 + z))   # model.py:143
    g_c0 = -np.sum(e)                                 # model.py:144
    grads = {"F": gF, "Psi": gPsi, "omega": gOm,
             "tau0": np.asarray(g_tau0, dtype=dtype), "c0": np.asarray(g_c0, dtype=dtype),
             "beta": np.asarray(g_beta, dtype=dtype)}
    if return_abs:
        # sum of |terms| of the three scalar sums: their condition (|sum| / sum|terms| is 1/50 .. 1/900 on data drawn
        # from the model); a float32 implementation is judged against 2^-24 x this, not against the cancelled sum
        absum = {"tau0": float(np.sum(np.abs(e * pw))), "c0": float(np.sum(np.abs(e))),
                 "beta": float(np.sum(np.abs(e * p["tau0"] * pw * np.log(1.0 + z))))}
        return nll, grads, absum
    return nll, grads


def forward(params, delta, error, zabs, mask, tau_which="becker", tau_series=1,
            dtype=np.float64, A_blue=None, return_sums=False):
    """Batch loss and normalised gradients (QFA/model.py:74-105).

    loss = mean NLL; grad[key] = sum_s g_s[key] / #{s : g_s[key] != 0}, elementwise;
    0/0 = NaN for pixels masked in every spectrum (quirk Q3).
    """
    B = len(delta)
    loss = 0.0
    sums, counts = None, None
    for s in range(B):
        nll, g = nll_and_grads_single(params, delta[s], error[s], zabs[s], mask[s],
                                      tau_which, tau_series, dtype,
                                      None if A_blue is None else A_blue[s])
        loss += nll / B
        if sums is None:
            sums = {k: np.zeros_like(v) for k, v in g.items()}
            counts = {k: np.zeros_like(v) for k, v in g.items()}
        for k in g:
            sums[k] = sums[k] + g[k]
            counts[k] = counts[k] + (g[k] != 0.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = {k: sums[k] / counts[k] for k in sums}
    if return_sums:
        return loss, grads, sums, counts
    return loss, grads


def predict_single(params, mu, flux, error, zabs, mask, tau_which="becker", tau_series=1,
                   dtype=np.float64, A_blue=None):
    """Posterior prediction (QFA/model.py:160-180).

    Returns ll (the NLL, as the reference names it), hmean (k,), hcov (k,k),
    cont (Npix,) = F hmean + mu on ALL pixels, unc (Npix,) = sqrt(diag(F hcov F^T)).
    """
    p = _as_params(params, dtype)
    F = p["F"]
    w = np.asarray(mask, dtype=bool)
    mu = np.asarray(mu, dtype=dtype)
    A, zdep, D = pixel_terms(params, error, zabs, tau_which, tau_series, dtype, A_blue)
    delta = np.where(w, np.asarray(flux, dtype=dtype) - mu * A, 0.0)   # model.py:166
    wD, d, M, C, y, u, nll = _lowrank_core(F, A, D, w, delta)
    hcov = np.linalg.inv(C)                                            # model.py:178
    hmean = y                                                          # model.py:179
    cont = F @ hmean + mu                                              # model.py:180
    unc = np.sqrt(np.einsum("ia,ab,ib->i", F, hcov, F))
    return nll, hmean, hcov, cont, unc


# --------------------------------------------------------------------------------------
# optimiser / parameter maintenance
# --------------------------------------------------------------------------------------

def step_lr(i, lr, alpha, step):
    """lr * alpha ** ((i+1)//step)  (QFA/optimizer.py:79-99)."""
    return lr * alpha ** ((i + 1) // step)


def adam_update(m, v, i, params, grads, lr, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-3,
                dtype=np.float64):
    """One Adam.update (QFA/optimizer.py:37-52): L2 folded into the gradient,
    bias-correction exponent i+1 where i only advances on Adam.step() (per epoch, quirk Q4).
    Returns (new_params, new_m, new_v)."""
    newp, newm, newv = {}, {}, {}
    for k in grads:
        g = np.asarray(grads[k], dtype=dtype) + weight_decay * np.asarray(params[k], dtype=dtype)
        newm[k] = (1 - b1) * g + b1 * np.asarray(m[k], dtype=dtype)
        newv[k] = (1 - b2) * g * g + b2 * np.asarray(v[k], dtype=dtype)
        mhat = newm[k] / (1.0 - b1 ** (i + 1))
        vhat = newv[k] / (1.0 - b2 ** (i + 1))
        newp[k] = np.asarray(params[k], dtype=dtype) - lr * mhat / (np.sqrt(vhat) + eps)
    return newp, newm, newv


def clip_params(params, min_value=1e-3, max_value=2.0):
    """QFA.clip (QFA/model.py:233-241)."""
    out = dict(params)
    out["omega"] = np.clip(params["omega"], min_value, max_value)
    out["Psi"] = np.clip(params["Psi"], min_value, max_value)
    out["tau0"] = np.clip(params["tau0"], 0.0, 1.0)
    out["beta"] = np.clip(params["beta"], 0.1, 5.0)
    out["c0"] = np.clip(params["c0"], -5.0, 5.0)
    return out


def _edge_mean(x, half):
    """Moving average over a (2*half+1) window along axis 0 whose divisor is the number of
    in-range samples (avg_pool with count_include_pad=False, QFA/model.py:243-252)."""
    x = np.asarray(x)
    n = x.shape[0]
    cs = np.concatenate([np.zeros((1,) + x.shape[1:], dtype=x.dtype), np.cumsum(x, axis=0)])
    idx = np.arange(n)
    lo = np.maximum(idx - half, 0)
    hi = np.minimum(idx + half + 1, n)
    cnt = (hi - lo).astype(x.dtype)
    return (cs[hi] - cs[lo]) / cnt.reshape((-1,) + (1,) * (x.ndim - 1))


def smooth_params(params):
    """QFA.smooth: 15-px window on omega and Psi, 31-px on F along pixels (model.py:243-252)."""
    out = dict(params)
    out["omega"] = _edge_mean(params["omega"], 7)
    out["Psi"] = _edge_mean(params["Psi"], 7)
    out["F"] = _edge_mean(params["F"], 15)
    return out


def woodbury_inverse(M, D):
    """Dense (MM^T + diag D)^-1 (QFA/utils.py:12-32); small sizes only."""
    M = np.asarray(M)
    D = np.asarray(D)
    k = M.shape[1]
    W = M / D[:, None]
    C = np.eye(k, dtype=M.dtype) + M.T @ W
    return np.diag(1.0 / D) - W @ np.linalg.solve(C, W.T)


def woodbury_logdet(M, D):
    """log det(MM^T + diag D) (QFA/utils.py:35-54), via Cholesky instead of det."""
    M = np.asarray(M)
    D = np.asarray(D)
    k = M.shape[1]
    C = np.eye(k, dtype=M.dtype) + M.T @ (M / D[:, None])
    return float(np.sum(np.log(D)) + 2.0 * np.sum(np.log(np.diag(np.linalg.cholesky(C)))))


def load_params_npz(path, quirk_c0_from_beta=True, dtype=np.float32):
    """QFA.load_from_npz (QFA/model.py:282-295) including quirk Q1: c0 <- file['beta']."""
    f = np.load(path)
    p = {k: np.asarray(f[k], dtype=dtype) for k in ("F", "Psi", "omega", "tau0", "beta")}
    p["c0"] = np.asarray(f["beta"] if quirk_c0_from_beta else f["c0"], dtype=dtype)
    mu = np.asarray(f["mu"], dtype=dtype)
    return p, mu


# --------------------------------------------------------------------------------------
# host preprocessing of the reference's dataloader
# --------------------------------------------------------------------------------------

def tau_total(wav_grid, zqso, which="becker"):
    """Total Lyman-series optical depth on the blue pixels (QFA/utils.py:174-203): every series
    line i whose wavelength exceeds wav_grid[0] contributes tau(z_i) on the pixels blueward of it,
    z_i = (1+zqso) wav / lambda_i - 1.  Returns (N, Nb), Nb = #(wav < 1215.6701)."""
    wav = np.asarray(wav_grid, dtype=np.float64)
    zq = np.asarray(zqso, dtype=np.float64).reshape(-1)
    level = int(np.sum(wav[0] < _LYMAN_LAM))          # lambdas decrease: the first `level` lines apply
    if level == 0:
        raise ValueError("Wavelength grid does not cover Lyman series lines")
    nb = int(np.sum(wav < _LYMAN_LAM[0]))
    out = np.zeros((len(zq), nb))
    for i in range(level):
        n_i = int(np.sum(wav < _LYMAN_LAM[i]))
        z_i = (zq + 1.0)[:, None] * wav[None, :n_i] / _LYMAN_LAM[i] - 1.0
        out[:, :n_i] += tau_eff(z_i, which, i + 1)
    return out


def boxcar_reflect(s, window_len=32):
    """numpy `smooth` of the reference (QFA/utils.py:206-219): reflect-pad, boxcar, trim."""
    s = np.asarray(s, dtype=np.float64)
    ext = np.r_[s[window_len - 1:0:-1], s, s[-2:-window_len - 1:-1]]
    y = np.convolve(np.ones(window_len) / window_len, ext, mode="valid")
    return y[int(window_len / 2 - 1):-int(window_len / 2)]


def zabs_from_zqso(wav_grid, zqso, nb):
    """(1+zqso) wav_blue / 1215.67 - 1 (QFA/dataloader.py:102)."""
    wav = np.asarray(wav_grid, dtype=np.float64)
    return (np.asarray(zqso, dtype=np.float64) + 1).reshape(-1, 1) * wav[:nb] / LYA - 1


def mu_estimate(wav_grid, flux, mask, zqso, nb, which="becker", window_len=16):
    """Mean continuum: sum_s flux * exp(+tau_total) * mask / #(flux != -999), then the boxcar
    (QFA/dataloader.py:110-112)."""
    flux = np.asarray(flux, dtype=np.float64)
    up = np.ones_like(flux)
    up[:, :nb] = np.exp(tau_total(wav_grid, zqso, which))
    raw = np.sum(flux * up * np.asarray(mask), axis=0) / np.sum(flux != -999.0, axis=0)
    return raw, boxcar_reflect(raw, window_len)


def delta_from_flux(wav_grid, flux, zqso, mu, nb, which="becker"):
    """delta = flux - mu * exp(-tau_total) (blue) / flux - mu (red) (QFA/dataloader.py:135-138)."""
    flux = np.asarray(flux, dtype=np.float64)
    dn = np.ones_like(flux)
    dn[:, :nb] = np.exp(-tau_total(wav_grid, zqso, which))
    return flux - np.asarray(mu, dtype=np.float64) * dn
