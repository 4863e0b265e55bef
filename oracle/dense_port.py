"""Dense O(N_pix^3) CPU port of the reference's per-spectrum step -- TEST/BENCH INFRASTRUCTURE.

This is the ``cpu_baseline`` ("kind": "port") timed by ``bench.py`` beside the GPU number:
a from-scratch torch-CPU float32 loop that issues the same *dense* operation sequence as the
reference (gather by mask, dense diag matrices, dense N x N Woodbury inverse, dense
N x N products for the F gradient), so it carries the reference's ~8 N_pix^3 flop cost.
It is validated against the imported reference in ``tests/golden/make_golden.py`` and against
the low-rank oracle in ``tests/test_oracle_golden.py``.  Never imported by ``qfa_amd``.

Reference lines followed: QFA/model.py:74-158 (forward + single-spectrum NLL/gradients),
QFA/utils.py:12-54 (MatrixInverse / MatrixLogDet), QFA/utils.py:57-106 (tauHI, omega_func,
becker tau).
"""
from __future__ import annotations

import torch

from .qfa_oracle import LOG2PI, LYMAN_COEFF, TAU_MODELS


def _tau(z, which, series):
    amp, scale, expo, offset = TAU_MODELS[which]
    return (amp * ((1.0 + z) * scale) ** expo + offset) * float(LYMAN_COEFF[series - 1])


def dense_single(P, delta, error, zabs, mask, which="becker", series=1):
    """One spectrum the expensive way.  P: dict of float32 torch tensors."""
    Nb_full = P["omega"].shape[0]
    mb = mask[:Nb_full]
    nb = int(mb.sum())
    nr = int(mask[Nb_full:].sum())
    n = nb + nr
    d = delta[mask]
    sig = error[mask]
    z = zabs[mb]
    one_r = torch.ones(nr, dtype=torch.float32)
    zero_r = torch.zeros(nr, dtype=torch.float32)
    # mean transmission and scaled loadings (model.py:125-127)
    a = torch.cat((torch.exp(-_tau(z, which, series)), one_r))
    Ad = torch.diag(a)
    M = Ad @ P["F"][mask]
    # diagonal of the covariance (model.py:128-131; utils.py:57-92)
    thi = P["tau0"] * (1.0 + z) ** P["beta"]
    zd = (1.0 - P["c0"] - torch.exp(-thi)) ** 2
    om = torch.cat((P["omega"][mb] * zd, zero_r))
    D = a * P["Psi"][mask] * a + om + sig * sig
    # dense Woodbury inverse and log-determinant (utils.py:29-32, 51-54)
    Dinv = torch.diag(1.0 / D)
    eye = torch.eye(M.shape[1], dtype=torch.float32)
    core = eye + M.T @ Dinv @ M
    Sinv = Dinv - Dinv @ M @ torch.linalg.inv(core) @ M.T @ Dinv
    logdet = torch.log(D).sum() + torch.log(torch.linalg.det(core))
    dc = d[:, None]
    nll = 0.5 * (dc.mT @ Sinv @ dc + n * LOG2PI + logdet)
    # dense derivative w.r.t. Sigma and the six parameter "gradients" (model.py:136-144)
    G = 0.5 * (Sinv - Sinv @ dc @ dc.mT @ Sinv)
    gF_m = 2 * Ad @ G @ Ad @ M
    g = torch.diag(G)
    gPsi_m = a * g * a
    gOm_m = g[:nb] * zd
    root = 1.0 - thi - P["c0"]
    e = g[:nb] * om[:nb] * zd * 2.0 * root
    pw = (1.0 + z) ** P["beta"]
    g_tau0 = -(e * pw).sum()
    g_beta = -(e * (P["tau0"] * pw * torch.log(1.0 + z))).sum()
    g_c0 = -e.sum()
    # scatter back to full length (model.py:145-150)
    gF = torch.zeros_like(P["F"], dtype=torch.float32)
    gF[mask] = gF_m
    gPsi = torch.zeros_like(P["Psi"], dtype=torch.float32)
    gPsi[mask] = gPsi_m
    gOm = torch.zeros_like(P["omega"], dtype=torch.float32)
    gOm[mb] = gOm_m
    return nll, {"F": gF, "Psi": gPsi, "omega": gOm, "tau0": g_tau0, "c0": g_c0, "beta": g_beta}


def dense_forward(P, delta, error, zabs, mask, which="becker", series=1):
    """Batch mean NLL and sum/count-normalised gradients (model.py:74-105)."""
    B = delta.shape[0]
    tot = None
    cnt = None
    loss = 0.0
    for s in range(B):
        nll, g = dense_single(P, delta[s], error[s], zabs[s], mask[s], which, series)
        loss = loss + nll / B
        if tot is None:
            tot = {k: torch.zeros_like(v) for k, v in g.items()}
            cnt = {k: torch.zeros_like(v) for k, v in g.items()}
        for k in g:
            tot[k] = tot[k] + g[k]
            cnt[k] = cnt[k] + (g[k] != 0.0)
    return loss, {k: tot[k] / cnt[k] for k in tot}


def to_torch_params(params):
    import numpy as np
    return {k: torch.tensor(np.asarray(params[k], dtype=np.float32)) for k in
            ("F", "Psi", "omega", "tau0", "c0", "beta")}
