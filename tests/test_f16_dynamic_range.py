"""Stage 1 of pass 2 and of the posterior writer runs on TWO float16 pieces per operand (qfa_amd/csrc/qfa_common.h "float16
pieces"): float16 has five exponent bits, so the image builders scale every pixel's F row and the kernels every spectrum's
[y | C^-1'] (writer: [hmean | hcov']) by powers of two.  The seeded shapes of the other parity tests keep F near 0.1 and the noise
near 0.1 -- nothing there would notice a scale that is wrong by 2^10.  Here F is stretched over many orders of magnitude (as a
whole, per pixel, per component) and the noise over two, and the float64 oracle (reference QFA/model.py:107-180 restated) is
the judge as everywhere else: any overflow of a float16 piece is an inf / NaN, any lost low piece an error of 1e-3."""
import numpy as np
import pytest

from conftest import rel_l2
from qfa_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def pixel_units(kind, npix):
    """A change of flux units per pixel: everything that carries flux (F, mu, flux, delta, sigma; Psi and omega squared) scales
    with it and the problem stays as well conditioned as the seeded one -- what moves is the magnitude of every operand."""
    if kind == "tiny":
        return np.full(npix, 3.0e-5)         # pair products ~1e-11: far below float16's smallest normal without the scale
    if kind == "huge":
        return np.full(npix, 3.0e3)          # F ~ 300: pair products ~1e5, above float16's largest (65 504) without it
    if kind == "pixel_ramp":
        return np.exp(np.linspace(np.log(1.0e-4), np.log(3.0e2), npix))
    return np.ones(npix)


def stretched(kind, npix, nb, nh, seed):
    """Parameters whose COMPONENTS span orders of magnitude (the data are then drawn from these parameters)."""
    from qfa_amd import synthetic
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=seed)
    rng = np.random.default_rng(4400 + seed)
    F = p["F"].astype(np.float64)
    if kind == "component_ramp":
        F *= (10.0 ** rng.uniform(-4.0, 0.0, size=nh))[None, :]
    elif kind == "one_line":                 # a component that lives on a dozen pixels, 1e4 x what it is elsewhere
        c = npix // 3
        F[:, 0] = 1.0e-4 * F[:, 0]
        F[c:c + 12, 0] = 0.3
    p["F"] = F.astype(np.float32)
    if kind == "psi_zero":                   # outside the reference's clip (Psi >= 1e-3): stage 3 of k_grads_t has no bound on beta there
        p["Psi"] = p["Psi"].copy()
        p["Psi"][::7] = 0.0
    return p, mu


KINDS = ("tiny", "huge", "pixel_ramp", "component_ramp", "one_line", "psi_zero")


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("nh,flag", [(8, 0), (8, "pixres"), (12, 0), (12, "pixres"), (16, "pixres"), (24, 0)])
def test_stretched_parameters_against_the_oracle(dev, kind, nh, flag):
    import torch
    from oracle import qfa_oracle as O
    from qfa_amd import QFA, synthetic
    npix, nb, B = 331, 140, 48
    seed = 17 * nh + KINDS.index(kind)
    wav = np.linspace(1216.0 - 180.0 * nb / npix - 1.0, 1216.0 + 300.0 * (npix - nb) / npix + 1.0, npix)
    p, mu = stretched(kind, npix, nb, nh, seed)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, B, seed=900 + seed, masks=True)
    # the noise over two orders of magnitude as well (weights 1 / D over four)
    rng = np.random.default_rng(5100 + seed)
    b["error"] = (b["error"] * 10.0 ** rng.uniform(-0.5, 1.5, size=b["error"].shape)).astype(np.float32)
    u = pixel_units(kind, npix)
    p = dict(p)
    p["F"] = (p["F"].astype(np.float64) * u[:, None]).astype(np.float32)
    p["Psi"] = (p["Psi"].astype(np.float64) * u * u).astype(np.float32)
    p["omega"] = (p["omega"].astype(np.float64) * (u * u)[:nb]).astype(np.float32)
    mu = (mu.astype(np.float64) * u).astype(np.float32)
    for k in ("delta", "error", "flux"):
        b[k] = (b[k].astype(np.float64) * u[None, :]).astype(np.float32)
    m = QFA(nb, npix - nb, nh, dev, model_params=p)
    m.mu = torch.as_tensor(mu, device=dev).to(torch.float32)
    m.flags = _lib.F_PASS2_PIXRES if flag == "pixres" else 0
    f32 = torch.float32
    d, e, z, mk = (torch.as_tensor(b[k], device=dev) for k in ("delta", "error", "zabs", "mask"))
    d, e, z, mk = d.to(f32), e.to(f32), z.to(f32).reshape(B, nb), mk.to(torch.bool)
    nll = torch.empty(B, dtype=f32, device=dev)
    acc = m.accumulate(d, e, z, mk, nll=nll)
    loss, gr = m._finalize(acc.clone(), True)
    oloss, ogr = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    case = (kind, nh, flag)
    ours_nll = nll.cpu().numpy()
    assert np.all(np.isfinite(ours_nll)), case
    per = np.array([O.nll_and_grads_single(p, b["delta"][s], b["error"][s], b["zabs"][s], b["mask"][s])[0] for s in range(B)])
    scale = np.maximum(np.abs(per), b["mask"].sum(axis=1))
    # (psi_zero is outside the reference's domain (clip: Psi >= 1e-3).  Pixels whose only variance is a small sigma^2 have weights
    # 1e3 x the others' and a leverage wD A^2 f^T C^-1 f near one: diag Sigma^-1 = wD (1 - leverage) cancels, and the 22 bits of the
    # float16 stage 1 show in gPsi at 5e-4 .. 3e-3 where the rest of this file stays under 1e-4.  The case is here for what it must
    # NOT do -- overflow a float16 piece into inf / NaN, or lose a digit of the NLL -- with bars ten to thirty times wider)
    hard = kind == "psi_zero"
    assert np.max(np.abs(ours_nll - per) / np.maximum(scale, 1.0)) < (5e-5 if hard else 1e-5), (case, np.max(np.abs(ours_nll - per) / np.maximum(scale, 1.0)))
    for k, tol in (("F", 3e-3 if hard else 3e-4), ("Psi", 5e-3 if hard else 1e-4), ("omega", 5e-3 if hard else 1e-4)):
        ours, ref = gr[k].cpu().numpy(), np.asarray(ogr[k])
        assert np.all(np.isfinite(ours)), (k, case)
        assert rel_l2(ours, ref) < tol, (k, case, rel_l2(ours, ref))
    # the posterior of the first rows through the writer (model.py:160-180)
    m.flags = 0
    fx = torch.as_tensor(b["flux"], device=dev).to(f32)
    pred = [x.cpu().numpy() for x in m.predict(fx, e, z, mk)]
    for s in range(4):
        o = O.predict_single(p, mu, b["flux"][s], b["error"][s], b["zabs"][s], b["mask"][s])
        assert np.all(np.isfinite(pred[3][s])) and np.all(np.isfinite(pred[4][s])), case
        # continuum within 1e-4 of the scale of the spectrum's continuum (BASELINE north_star: posterior continua 1e-4 relative)
        cs = np.max(np.abs(o[3]))
        assert np.max(np.abs(pred[3][s] - o[3])) < (5e-4 if hard else 1e-4) * cs, ("cont", case, np.max(np.abs(pred[3][s] - o[3])) / cs)
        assert rel_l2(pred[4][s], o[4]) < (1e-3 if hard else 2e-4), ("unc", case, rel_l2(pred[4][s], o[4]))
