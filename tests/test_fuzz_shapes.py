"""Seeded random shapes through every input form and pass-2 form against the float64 oracle.

The fixed cases of test_hip_parity.py / test_resident_form.py pin the layouts someone thought of (ragged tiles, the blue / red
boundary inside a tile, N_h on both sides of 8 / 16, ...).  This file draws the shape instead: N_pix 1..1600 (log-uniform),
N_b anywhere in 0..N_pix, N_h 1..32, 1..160 spectra, masks with runs and a fully masked spectrum now and then, the pass-2 form
(default dispatch, pixel-resident, two-role XDL, float32 MFMA) and the input form (zabs tensors, factored-z tensors, resident rows
picked by a permutation) -- the reference's own loop restated in oracle/qfa_oracle.py (model.py:74-158) is the judge of each;
then the posterior of the same rows (model.py:160-180) through the same input form.
"""
import os

import numpy as np
import pytest

from conftest import rel_l2
from qfa_amd import _lib

pytestmark = pytest.mark.gpu

KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")
TOL_NLL = 5e-6
TOL_G = {"F": 1e-4, "Psi": 2e-5, "omega": 2e-5}      # (the scalar gradients: in units of the sum of |terms|, below)


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def draw_case(seed):
    rng = np.random.default_rng(770000 + seed)
    npix = int(np.exp(rng.uniform(0.0, np.log(1600.0))))
    npix = max(1, npix)
    edge = rng.integers(0, 6)
    nb = {0: 0, 1: npix, 2: max(0, npix - 1), 3: min(npix, 16 * int(rng.integers(0, npix // 16 + 1)))}.get(int(edge), int(rng.integers(0, npix + 1)))
    nh = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 12, 15, 16, 17, 20, 24, 31, 32]))
    B = int(np.exp(rng.uniform(0.0, np.log(160.0))))
    forms = ["zabs", "zfac", "rows"]
    form = forms[int(rng.integers(0, 3))]
    flags = [0, _lib.F_PASS2_XDL, _lib.F_PASS2_F32] + ([_lib.F_PASS2_PIXRES] * 2 if nh <= 16 else [])
    flag = int(flags[int(rng.integers(0, len(flags)))])
    return npix, nb, nh, max(1, B), form, flag, rng


# (QFA_FUZZ_CASES=2000 python -m pytest tests/test_fuzz_shapes.py -m gpu: the wider sweep run by hand, 50 s)
@pytest.mark.parametrize("seed", range(int(os.environ.get("QFA_FUZZ_CASES", "96"))))
def test_random_shape_against_the_oracle(dev, seed):
    import torch
    from oracle import qfa_oracle as O
    from qfa_amd import QFA, synthetic
    from qfa_amd.resident import ResidentBatch
    npix, nb, nh, B, form, flag, rng = draw_case(seed)
    wav = np.linspace(1216.0 - 180.0 * nb / max(npix, 1) - 1.0, 1216.0 + 300.0 * (npix - nb) / max(npix, 1) + 1.0, npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=seed)
    N = B + int(rng.integers(0, 40)) if form == "rows" else B
    b = synthetic.make_batch_numpy(p, mu, wav, nb, N, seed=1000 + seed, masks=True)
    if N > 2 and rng.random() < 0.3:
        b["mask"][int(rng.integers(0, N)), :] = False                       # a spectrum without a valid pixel
    m = QFA(nb, npix - nb, nh, dev, model_params=p)
    m.mu = torch.as_tensor(mu, device=dev).to(torch.float32)
    m.flags = flag
    f32 = torch.float32
    d, e, mk = (torch.as_tensor(b[k], device=dev) for k in ("delta", "error", "mask"))
    d, e, mk = d.to(f32), e.to(f32), mk.to(torch.bool)
    zq1 = torch.as_tensor((1.0 + b["zqso"].astype(np.float64)).astype(np.float32), device=dev)
    ratio = torch.as_tensor((wav[:nb] / synthetic.LYA).astype(np.float32), device=dev)
    rows = np.arange(B)
    nll = torch.empty(B, dtype=f32, device=dev)
    if form == "zabs":
        z = torch.as_tensor(b["zabs"], device=dev).to(f32).reshape(B, nb)
        acc = m.accumulate(d, e, z, mk, nll=nll)
    elif form == "zfac":
        acc = m.accumulate(d, e, None, mk, nll=nll, zfac=(zq1, ratio))
    else:
        rows = rng.permutation(N)[:B]
        stride = (npix + 31) // 32 * 32

        def pad(t, fill):
            out = torch.full((N, stride), fill, dtype=t.dtype, device=dev)
            out[:, :npix] = t
            return out
        rb = ResidentBatch(None, pad(d, -7.0e9), pad(e, float("nan")), pad(mk, True), zq1, ratio,
                           torch.as_tensor(rows.astype(np.int32), device=dev), npix, nb)
        acc = m.accumulate(batch=rb, nll=nll)
    loss, gr = m._finalize(acc.clone(), True)
    # the posterior of the same rows (reference QFA/model.py:160-180): raw flux in, five outputs
    fx = torch.as_tensor(b["flux"], device=dev).to(f32)
    m.flags = 0
    if form == "zabs":
        pred = m.predict(fx, e, z, mk)
    elif form == "zfac":
        pred = m.predict(fx, e, None, mk, zfac=(zq1, ratio))
    else:
        rb.flux = pad(fx, 3.0e9)
        pred = m.predict(batch=rb)
    pred = [x.cpu().numpy() for x in pred]
    sel = {k: b[k][rows] for k in ("delta", "error", "zabs", "mask")}
    oloss, ogr = O.forward(p, sel["delta"], sel["error"], sel["zabs"], sel["mask"])
    per = np.empty(B)
    absum, cnt = {k: 0.0 for k in ("tau0", "c0", "beta")}, {k: 0 for k in ("tau0", "c0", "beta")}
    for s in range(B):
        per[s], g1, ab = O.nll_and_grads_single(p, sel["delta"][s], sel["error"][s], sel["zabs"][s], sel["mask"][s], return_abs=True)
        for k in absum:
            absum[k] += ab[k]
            cnt[k] += int(g1[k] != 0.0)
    case = (npix, nb, nh, B, form, flag)
    ours_nll = nll.cpu().numpy()
    assert np.all(np.isfinite(ours_nll)), case
    # (an NLL is a sum of terms of order one per unmasked pixel that may cancel on short spectra: relative to the larger of |NLL|
    # and the pixel count, as in test_blue_red_boundary_layouts)
    scale = np.maximum(np.abs(per), sel["mask"].sum(axis=1))
    assert np.max(np.abs(ours_nll - per) / np.maximum(scale, 1.0)) < TOL_NLL, case
    assert abs(loss.item() - oloss) <= TOL_NLL * max(abs(oloss), float(scale.mean())), case
    for k in KEYS:
        ours, ref = gr[k].cpu().numpy(), np.asarray(ogr[k])
        assert np.array_equal(np.isnan(ours), np.isnan(ref)), (k, case)
        ok = ~np.isnan(ref)
        if k in absum:
            # sums of terms that cancel (here up to several 1 000x: a dozen short spectra).  In units of the sum of |terms|:
            # large sums average the float32 error of their terms down to the 1.5e-7 that test_full_size_parity.py asserts; a
            # sum of a few terms shows the error of one term -- a chain of ~10 float32 operations with three hardware
            # transcendentals, a few 1e-7 to 4e-6 over 3 000 drawn shapes.  A wrong term is an error of order 1e-2 .. 1.
            if ok.all() and absum[k] > 0:
                assert abs(float(ours) - float(ref)) * cnt[k] <= 1e-5 * absum[k], (k, case, abs(float(ours) - float(ref)) * cnt[k] / absum[k])
        elif ok.any() and np.linalg.norm(ref[ok]) > 0:
            # (a gradient vector of one or two elements over a handful of spectra is itself a cancelling sum: the relative
            # bars of the realistic shapes apply from 8 elements and 4 spectra on, 1e-3 below that)
            tol = TOL_G[k] if (ok.sum() >= 8 and B >= 4) else 1e-3
            assert rel_l2(ours[ok], ref[ok]) < tol, (k, case, rel_l2(ours[ok], ref[ok]))
    for s_ in range(min(B, 3)):
        r = rows[s_]
        o = O.predict_single(p, mu, b["flux"][r], b["error"][r], b["zabs"][r], b["mask"][r])
        nvalid = int(b["mask"][r].sum())
        assert abs(pred[0][s_] - o[0]) <= TOL_NLL * max(abs(o[0]), nvalid, 1.0), ("ll", case)
        if nvalid == 0:
            continue                                        # (no data: the posterior is the prior, compared through ll above)
        assert rel_l2(pred[1][s_], o[1]) < 2e-4, ("hmean", case, rel_l2(pred[1][s_], o[1]))
        assert rel_l2(pred[2][s_], o[2]) < 2e-4, ("hcov", case, rel_l2(pred[2][s_], o[2]))
        assert np.max(np.abs(pred[3][s_] - o[3])) <= 1e-4 * np.max(np.abs(o[3])), ("cont", case)     # north_star: 1e-4
        assert rel_l2(pred[4][s_], o[4]) < 1e-4, ("unc", case, rel_l2(pred[4][s_], o[4]))
