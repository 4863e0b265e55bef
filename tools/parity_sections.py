"""Section-by-section parity of the packed accumulation buffer at BASELINE's full sizes (GPU box only).

The packed buffer of ``qfa_nll_grad_f32`` is  [accF | sumA | gPsi | gOmega | cnt | g_tau0 g_c0 g_beta n_blue sumNLL n 0 0]
(include/qfa_hip.h).  ``section_errors`` runs one big launch over the whole batch and compares EACH section with a
float64 sum of the same batch accumulated in small HIP chunks (<= 512 spectra: the small-batch path, itself checked
against the oracle by tests/test_hip_parity.py) -- so a wrong scalar gradient at scale cannot hide behind the 64 000
floats of accF.  ``oracle_subbatch_errors`` compares the normalised gradients of a sampled sub-batch with the
float64 CPU oracle.  Used by tests/test_full_size_parity.py (asserts) and tools/accuracy_report.py (prints).
"""
import numpy as np
import torch

KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")
SCALARS = ("g_tau0", "g_c0", "g_beta", "n_blue", "sum_nll", "n_spectra")


def sections(model):
    npix, nb, nh = model.Npix, model.Nb, model.Nh
    o = [0, npix * nh]
    for n in (npix, npix, nb, npix):
        o.append(o[-1] + n)
    names = ("accF", "sumA", "gPsi", "gOmega", "cnt")
    sl = {n: slice(o[i], o[i + 1]) for i, n in enumerate(names)}
    for j, n in enumerate(SCALARS):
        sl[n] = slice(o[-1] + j, o[-1] + j + 1)
    return sl


def make_config_batch(p, mu, wav, nb, B, seed, dev, masks, slab=25000):
    from qfa_amd import synthetic
    parts = [synthetic.make_batch_torch(p, mu, wav, nb, min(slab, B - s0), seed + 17 * i, dev, masks=masks)
             for i, s0 in enumerate(range(0, B, slab))]
    out = tuple(torch.cat([q[j] for q in parts]) for j in range(4))
    del parts
    torch.cuda.empty_cache()
    return out


def chunked_f64(model, batch, chunk=512):
    """float64 sum of the packed buffers of <= chunk-spectrum launches; per-spectrum NLL of those launches"""
    d, e, z, mk = batch
    B = d.shape[0]
    tot = None
    nll = torch.empty(B, dtype=torch.float32, device=d.device)
    for a in range(0, B, chunk):
        b = min(a + chunk, B)
        acc = model.accumulate(d[a:b], e[a:b], z[a:b], mk[a:b], nll=nll[a:b])
        tot = acc.double() if tot is None else tot + acc.double()
    return tot, nll


def section_errors(model, batch, chunk=512):
    """{section: error} of ONE launch over the whole batch against the chunked float64 sum.  Vector sections:
    relative L2; scalars: |a - b| / |b|; counts must be exact (error 0 or 1)."""
    d, e, z, mk = batch
    B = d.shape[0]
    nll = torch.empty(B, dtype=torch.float32, device=d.device)
    big = model.accumulate(d, e, z, mk, nll=nll).double().cpu().numpy()
    ref, nll_c = chunked_f64(model, batch, chunk)
    ref = ref.cpu().numpy()
    out = {}
    for name, sl in sections(model).items():
        a, b = big[sl], ref[sl]
        if name in ("cnt", "n_blue", "n_spectra"):
            out[name] = float(np.max(np.abs(a - b)))
        elif a.size == 1:
            out[name] = float(abs(a[0] - b[0]) / max(abs(b[0]), 1e-300))
        else:
            out[name] = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    n1, n2 = nll.cpu().numpy().astype(np.float64), nll_c.cpu().numpy().astype(np.float64)
    out["nll_per_spectrum_max_rel"] = float(np.max(np.abs(n1 - n2) / np.abs(n2)))      # worst of B spectra
    out["nll_per_spectrum_rel_l2"] = float(np.linalg.norm(n1 - n2) / np.linalg.norm(n2))
    out["finite"] = bool(np.isfinite(big).all())
    return out


def oracle_subbatch_errors(model, p, batch, idx):
    """normalised gradients + loss + per-spectrum NLL of the sub-batch `idx` (its own launch) vs the float64 oracle.
    Keys "<k>_np32": the same error for the oracle evaluated in float32 numpy.  The three scalar gradients are sums of
    strongly cancelling terms (data drawn from the model itself: expectation zero): "<k>_over_abs" is their error divided
    by the sum of |terms| (what a float32 implementation can be held to: a few 2^-24), "<k>_cancellation" the ratio
    sum|terms| / |sum|."""
    from oracle import qfa_oracle as O
    d, e, z, mk = (x[idx] for x in batch)
    n = d.shape[0]
    nll = torch.empty(n, dtype=torch.float32, device=d.device)
    acc = model.accumulate(d, e, z, mk, nll=nll)
    loss, gr = model._finalize(acc, True)
    dn, en, zn, mn = (x.cpu().numpy() for x in (d, e, z, mk))
    per = np.empty(n)
    sums = counts = sums32 = None
    absum = {"tau0": 0.0, "c0": 0.0, "beta": 0.0}
    for s in range(n):
        per[s], g, ab = O.nll_and_grads_single(p, dn[s], en[s], zn[s], mn[s], return_abs=True)
        for k in absum:
            absum[k] += ab[k]
        _, g32 = O.nll_and_grads_single(p, dn[s], en[s], zn[s], mn[s], dtype=np.float32)
        if sums is None:
            sums = {k: np.zeros_like(v) for k, v in g.items()}
            counts = {k: np.zeros_like(v) for k, v in g.items()}
            sums32 = {k: np.zeros_like(v) for k, v in g32.items()}
        for k in g:
            sums[k] += g[k]
            counts[k] += (g[k] != 0.0)
            sums32[k] += g32[k]
    out = {"loss": float(abs(loss.item() - per.mean()) / abs(per.mean())),
           "nll_per_spectrum_max_rel": float(np.max(np.abs(nll.cpu().numpy() - per) / np.abs(per)))}
    with np.errstate(invalid="ignore", divide="ignore"):
        for k in KEYS:
            ref = sums[k] / counts[k]
            ours = gr[k].cpu().numpy().astype(np.float64)
            ok = ~np.isnan(ref)
            out["nan_pattern_" + k] = bool(np.array_equal(np.isnan(ours), np.isnan(ref)))
            out[k] = float(np.linalg.norm(ours[ok] - ref[ok]) / max(np.linalg.norm(ref[ok]), 1e-300))
            r32 = (sums32[k] / counts[k]).astype(np.float64)
            out[k + "_np32"] = float(np.linalg.norm(r32[ok] - ref[ok]) / max(np.linalg.norm(ref[ok]), 1e-300))
        for k in ("tau0", "c0", "beta"):                # error in units of sum|terms| (condition-free), and the condition
            ours = float(gr[k].item())
            ref = float(sums[k] / counts[k])
            out[k + "_over_abs"] = abs(ours - ref) * float(counts[k]) / max(absum[k], 1e-300)
            out[k + "_cancellation"] = absum[k] / max(abs(float(sums[k])), 1e-300)
    return out
