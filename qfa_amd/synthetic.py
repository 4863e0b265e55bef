"""Synthetic spectra drawn from QFA's own generative model.

Used by the benchmark, the smoke test and the parity tests (same tensors on both sides).
The recipe is the one fixed in SURVEY.md section 8(d):

* wavelength grid  ``10**arange(log10 1030, log10 1600, 1e-4)`` (1913 px, reference
  QFA/dataloader.py:61-63, QFA/config.py:38-40) or ``10**linspace(..., N_pix, endpoint=False)``;
* ``z_qso ~ U(2, 3.5)`` (QFA/config.py:35-36), ``zabs = (1+z) lambda_blue / 1215.67 - 1``
  (QFA/dataloader.py:102);
* ``flux = A (mu + F h + sqrt(Psi) e1) + sqrt(omega zdep) e2 + sigma e3``,
  ``delta = flux - mu A`` (reference README.md:46-54, QFA/model.py:125-131);
* masks: per spectrum Poisson(2) runs of length U{10..100} plus 1 % iid dropped pixels,
  masked flux/error set to the -999 sentinel (QFA/dataloader.py:24-28).

Two generators: numpy (bit-stable across machines, used for golden fixtures and tests) and
torch (on-device, used for benchmark-sized batches).
"""
from __future__ import annotations

import math

import numpy as np

LYA = 1215.67
_BECKER = (0.751, 1.0 / 4.5, 2.90, -0.132)


def wavelength_grid(n_pix=None):
    lo, hi = math.log10(1030.0), math.log10(1600.0)
    if n_pix is None:
        wav = 10 ** np.arange(lo, hi, 1e-4)
    else:
        wav = 10 ** np.linspace(lo, hi, int(n_pix), endpoint=False)
    nb = int(np.sum(wav < LYA))
    return wav, nb, len(wav) - nb


def desi_grid():
    """A rest-frame grid with the pixel counts of the reference's second shipped model, data/model_parameters_desi.npz
    (N_pix = 9243, N_b = 2238): linear, 1040 A + 0.0785 A per pixel (the file does not record its own grid)."""
    wav = 1040.0 + (LYA - 1040.0) / 2237.5 * np.arange(9243)
    nb = int(np.sum(wav < LYA))
    return wav, nb, len(wav) - nb


def mock_parameters(n_pix, nb, nh, seed=0, mu=None):
    """Smooth random loadings and realistic noise scales (SURVEY 8(d) 'Parameters')."""
    rng = np.random.default_rng(seed)
    F = 0.1 * rng.standard_normal((n_pix, nh))
    kw = 31 if n_pix >= 31 else max(1, n_pix - 1 + n_pix % 2)      # (odd, no longer than the axis: 'same' keeps n_pix rows)
    ker = np.ones(kw) / kw
    F = np.stack([np.convolve(F[:, a], ker, mode="same") for a in range(nh)], axis=1) * math.sqrt(kw)
    x = np.linspace(0.0, 1.0, n_pix)
    if mu is None:
        mu = 1.0 + 0.5 * np.exp(-0.5 * ((x - 0.32) / 0.03) ** 2) + 0.3 * np.exp(-0.5 * ((x - 0.9) / 0.05) ** 2)
    params = {
        "F": F.astype(np.float32),
        "Psi": np.full(n_pix, 0.1, dtype=np.float32) * (0.5 + rng.random(n_pix)).astype(np.float32),
        "omega": np.full(nb, 0.25, dtype=np.float32) * (0.5 + rng.random(nb)).astype(np.float32),
        "tau0": np.float32(0.146),
        "beta": np.float32(1.333),
        "c0": np.float32(0.239),
    }
    return params, np.asarray(mu, dtype=np.float32)


def _tau_becker(z):
    a, s, e, c = _BECKER
    return a * ((1.0 + z) * s) ** e + c


def make_batch_numpy(params, mu, wav, nb, batch, seed, masks=True, red_only=(), dead_range=None):
    """Returns dict(delta, error, zabs, mask, flux, zqso) as float32/bool numpy arrays."""
    rng = np.random.default_rng(seed)
    n_pix = len(wav)
    F = np.asarray(params["F"], dtype=np.float64)
    Psi = np.asarray(params["Psi"], dtype=np.float64)
    om = np.asarray(params["omega"], dtype=np.float64)
    tau0, beta, c0 = float(params["tau0"]), float(params["beta"]), float(params["c0"])
    mu = np.asarray(mu, dtype=np.float64)
    zq = rng.uniform(2.0, 3.5, size=batch)
    zabs = (1.0 + zq)[:, None] * wav[None, :nb] / LYA - 1.0
    A = np.ones((batch, n_pix))
    A[:, :nb] = np.exp(-_tau_becker(zabs))
    h = rng.standard_normal((batch, F.shape[1]))
    snr = np.exp(rng.uniform(math.log(2.0), math.log(100.0), size=batch))
    sigma = mu[None, :] / snr[:, None] * (0.8 + 0.4 * rng.random((batch, n_pix)))
    zdep = (1.0 - c0 - np.exp(-tau0 * (1.0 + zabs) ** beta)) ** 2
    flux = A * (mu[None, :] + h @ F.T + np.sqrt(Psi)[None, :] * rng.standard_normal((batch, n_pix)))
    flux[:, :nb] += np.sqrt(om[None, :] * zdep) * rng.standard_normal((batch, nb))
    flux += sigma * rng.standard_normal((batch, n_pix))
    mask = np.ones((batch, n_pix), dtype=bool)
    if masks:
        for s in range(batch):
            for _ in range(rng.poisson(2.0)):
                ln = int(rng.integers(10, 101))
                st = int(rng.integers(0, n_pix))
                mask[s, st:st + ln] = False
        mask &= rng.random((batch, n_pix)) >= 0.01
    for s in red_only:
        mask[s, :nb] = False
    if dead_range is not None:
        mask[:, dead_range[0]:dead_range[1]] = False
    flux = np.where(mask, flux, -999.0)
    sigma = np.where(mask, sigma, -999.0)
    delta = flux - mu[None, :] * A
    return {
        "delta": delta.astype(np.float32), "error": sigma.astype(np.float32),
        "zabs": zabs.astype(np.float32), "mask": mask, "flux": flux.astype(np.float32),
        "zqso": zq.astype(np.float32),
    }


def make_batch_torch(params, mu, wav, nb, batch, seed, device, masks=True, return_zq=False, return_flux=False):
    """On-device generator for benchmark-sized batches (same model, torch RNG).

    Returns (delta, error, zabs, mask) float32/bool tensors on ``device``; with ``return_zq`` also z_qso (batch,); with
    ``return_flux`` (flux, error, z_qso) -- what a loader is given (masked pixels hold -999 in both arrays).
    """
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    f32 = dict(dtype=torch.float32, device=device)
    n_pix = len(wav)
    F = torch.as_tensor(np.asarray(params["F"], dtype=np.float32), device=device)
    Psi = torch.as_tensor(np.asarray(params["Psi"], dtype=np.float32), device=device)
    om = torch.as_tensor(np.asarray(params["omega"], dtype=np.float32), device=device)
    mu_t = torch.as_tensor(np.asarray(mu, dtype=np.float32), device=device)
    wav_b = torch.as_tensor(np.asarray(wav[:nb], dtype=np.float32), device=device)
    tau0, beta, c0 = float(params["tau0"]), float(params["beta"]), float(params["c0"])
    zq = 2.0 + 1.5 * torch.rand(batch, generator=g, **f32)
    zabs = (1.0 + zq)[:, None] * wav_b[None, :] / LYA - 1.0
    a, s, e, c = _BECKER
    A = torch.ones(batch, n_pix, **f32)
    A[:, :nb] = torch.exp(-(a * ((1.0 + zabs) * s) ** e + c))
    h = torch.randn(batch, F.shape[1], generator=g, **f32)
    snr = torch.exp(math.log(2.0) + (math.log(100.0) - math.log(2.0)) * torch.rand(batch, generator=g, **f32))
    sigma = mu_t[None, :] / snr[:, None] * (0.8 + 0.4 * torch.rand(batch, n_pix, generator=g, **f32))
    flux = mu_t[None, :] + h @ F.T
    flux += torch.sqrt(Psi)[None, :] * torch.randn(batch, n_pix, generator=g, **f32)
    flux *= A
    zdep = (1.0 - c0 - torch.exp(-tau0 * (1.0 + zabs) ** beta)) ** 2
    flux[:, :nb] += torch.sqrt(om[None, :] * zdep) * torch.randn(batch, nb, generator=g, **f32)
    del zdep
    flux += sigma * torch.randn(batch, n_pix, generator=g, **f32)
    mask = torch.ones(batch, n_pix, dtype=torch.bool, device=device)
    if masks:
        # Poisson(2) runs of length U{10..100}: draw a fixed 8 candidate runs per spectrum and
        # keep the first n_s ~ Poisson(2) of them (P[n > 8] < 3e-4, truncated).
        n_runs = torch.poisson(torch.full((batch,), 2.0, **f32), generator=g).clamp_(max=8)
        start = torch.randint(0, n_pix, (batch, 8), generator=g, device=device)
        length = torch.randint(10, 101, (batch, 8), generator=g, device=device)
        pix = torch.arange(n_pix, device=device)[None, :]
        for r in range(8):
            live = (n_runs > r)[:, None]
            mask &= ~(live & (pix >= start[:, r:r + 1]) & (pix < (start[:, r:r + 1] + length[:, r:r + 1])))
        mask &= torch.rand(batch, n_pix, generator=g, **f32) >= 0.01
        flux = torch.where(mask, flux, torch.full_like(flux, -999.0))
        sigma = torch.where(mask, sigma, torch.full_like(sigma, -999.0))
    if return_flux:
        return flux.contiguous(), sigma.contiguous(), zq
    delta = flux - mu_t[None, :] * A
    if return_zq:
        return delta.contiguous(), sigma.contiguous(), zabs.contiguous(), mask.contiguous(), zq
    return delta.contiguous(), sigma.contiguous(), zabs.contiguous(), mask.contiguous()
