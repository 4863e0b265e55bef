"""Stage 3 of pass 2 (the F-gradient contraction G_s = F_tile Z_s, summed over K and over every spectrum of the batch)
under conditions chosen to expose a narrow product: F columns spanning 10^3 in scale, parameters after 200 training
steps (near-stationary: the gradient is a small difference of large sums), and the NORMALISED gradients of a 20 000-
spectrum batch of the headline shape against the float64 oracle on the host cores -- HIP against the oracle, not HIP
against HIP.  The default build issues six bf16 piece products per float32 product (float32 grade);
flags = F_S3_FAST selects three (operands carried to ~17 bits) and is measured beside it.  Reference semantics:
QFA/model.py:100-104,136-137 (sum over spectra / count), :204-215 (the training loop)."""
import numpy as np
import pytest

from conftest import rel_l2
from qfa_amd import _lib

pytestmark = pytest.mark.gpu
KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _model(dev, p, mu, nb, nr, nh, flags=0):
    import torch
    from qfa_amd import QFA
    m = QFA(nb, nr, nh, dev, model_params=p)
    m.mu = torch.tensor(mu, device=dev)
    m.flags = flags
    return m


def _bt(b, dev):
    import torch
    return tuple(torch.tensor(b[k], device=dev) for k in ("delta", "error", "zabs", "mask"))


def _gF_terms_scale(m, acc):
    """per-element size of the two sums whose difference is gF = F sumA - accF (the yardstick in the cancellation regime)"""
    n = m.Npix * m.Nh
    accF = acc[:n].double().cpu().numpy().reshape(m.Npix, m.Nh)
    cnt = acc[n + 2 * m.Npix + m.Nb: n + 3 * m.Npix + m.Nb].double().cpu().numpy()
    return np.linalg.norm(accF / np.maximum(cnt, 1.0)[:, None])


@pytest.mark.parametrize("npix,nh,B", [(640, 16, 48), (640, 32, 40), (352, 12, 70)])
def test_f_columns_spanning_three_decades(dev, npix, nh, B):
    """F columns scaled by 10^-1.5 .. 10^+1.5 (the spectra are drawn from THAT model): G = F Z then has products of very
    different size under one accumulator.  Six-term default within the fixed tolerances; the three-term form is looser."""
    from oracle import qfa_oracle as O
    from qfa_amd import synthetic
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=900 + nh)
    p = dict(p)
    p["F"] = (p["F"] * (10.0 ** np.linspace(-1.5, 1.5, nh))[None, :]).astype(np.float32)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, B, seed=901 + nh)
    ol, og = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    errs = {}
    # (QFA_F_S3_FAST: N_h = 17..32 only in the shipped library)
    forms = (("six", 0),) + ((("fast", _lib.F_S3_FAST),) if nh > 16 else (("pixres", _lib.F_PASS2_PIXRES),))
    for name, fl in forms:
        m = _model(dev, p, mu, nb, nr, nh, fl)
        loss, g = m.forward(*_bt(b, dev))
        assert abs(loss.item() - ol) <= 2e-5 * abs(ol)          # (C spans 10^6 here; 7e-6 achieved against 2e-7 on plain data)
        errs[name] = {k: rel_l2(g[k].cpu().numpy(), og[k]) for k in KEYS}
        # column by column: a small column must not drown in the error of a large one
        gf, rf = g["F"].cpu().numpy().astype(np.float64), og["F"]
        errs[name]["F_worst_column"] = max(rel_l2(gf[:, a], rf[:, a]) for a in range(nh))
    print("F columns over 10^3:", npix, nh, errs)
    six = errs["six"]
    # achieved (profiles/r3_accuracy.txt): F 2e-5 .. 1.4e-4, worst column 1.8e-4 -- the same with six and with three
    # products: at B <= 70 the error of this case is pass 1's (moments spanning 10^6), not stage 3's
    # (N_h = 32: F 7e-4, worst column 1.1e-3 -- equal to the last digit with four and with three products: with 32 columns
    # over 10^3 the k x k system C spans 10^6 and its float32 moments set the error, not the F contraction)
    lim = 2e-3 if nh > 16 else 3e-4
    assert six["F"] < lim and six["F_worst_column"] < 2 * lim, six
    assert six["Psi"] < 2e-5 and six["omega"] < 2e-5, six
    if "fast" in errs:
        assert errs["fast"]["F"] < 3e-3, errs["fast"]            # (not the default)
    if "pixres" in errs:                                         # the pixel-resident form of pass 2 (k_grads_t): same bar as six
        t = errs["pixres"]
        assert t["F"] < lim and t["F_worst_column"] < 2 * lim and t["Psi"] < 2e-5 and t["omega"] < 2e-5, t


@pytest.mark.parametrize("nh", [16, 32])
def test_near_stationary_parameters_after_200_steps(dev, nh):
    """200 Adam steps on a fixed data set, then the gradient of a batch at THOSE parameters: the F gradient is now a small
    difference of F sumA and accF.  Error measured against the float64 oracle, relative to the gradient AND to the size of
    the cancelling sums."""
    from oracle import qfa_oracle as O
    from qfa_amd import Adam, step_scheduler, synthetic
    npix, N, B = 640, 512, 128
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=40 + nh)
    data = synthetic.make_batch_numpy(p, mu, wav, nb, N, seed=41 + nh)
    p0 = dict(p)
    p0["F"] = (p["F"] * 0.7 + 0.03 * np.random.default_rng(5).standard_normal(p["F"].shape)).astype(np.float32)
    m = _model(dev, p0, mu, nb, nr, nh)
    opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 20), learning_rate=2e-3, weight_decay=1e-3)
    bt = _bt(data, dev)
    for step in range(200):
        a = (step % (N // B)) * B
        m.step(opt, *(x[a:a + B] for x in bt))
        if (step + 1) % (N // B) == 0:
            opt.step()
    trained = {k: m.parameters[k].cpu().numpy() for k in KEYS}
    assert all(np.isfinite(v).all() for v in trained.values())
    sub = {k: data[k][:96] for k in ("delta", "error", "zabs", "mask")}
    ol, og = O.forward(trained, sub["delta"], sub["error"], sub["zabs"], sub["mask"])
    out = {}
    for name, fl in (("six", 0),) + ((("pixres", _lib.F_PASS2_PIXRES),) if nh == 16 else (("fast", _lib.F_S3_FAST),)):
        mm = _model(dev, trained, mu, nb, nr, nh, fl)
        acc = mm.accumulate(*_bt(sub, dev)).clone()
        loss, g = mm._finalize(acc, True)
        d = g["F"].cpu().numpy().astype(np.float64) - og["F"]
        out[name] = {"F_rel": rel_l2(g["F"].cpu().numpy(), og["F"]), "F_over_terms": np.linalg.norm(d) / _gF_terms_scale(mm, acc),
                     "Psi": rel_l2(g["Psi"].cpu().numpy(), og["Psi"]), "omega": rel_l2(g["omega"].cpu().numpy(), og["omega"]),
                     "cancellation": _gF_terms_scale(mm, acc) / np.linalg.norm(og["F"])}
    print("after 200 steps:", nh, out)
    six = out["six"]
    # achieved: F_rel 5.4e-6 / 8.6e-6 (N_h = 16 / 32) with the default (six products everywhere from round 4 on); 8.6e-6 with
    # F_S3_FAST at N_h = 32
    assert six["F_over_terms"] < 8e-6 and six["F_rel"] < 3e-5, six
    assert six["Psi"] < 2e-5 and six["omega"] < 2e-5, six
    if "fast" in out:
        assert out["fast"]["F_rel"] < 2e-4, out["fast"]
    if "pixres" in out:                                          # k_grads_t: the same bars as the six-term default
        t = out["pixres"]
        assert t["F_over_terms"] < 8e-6 and t["F_rel"] < 3e-5 and t["Psi"] < 2e-5 and t["omega"] < 2e-5, t


def test_headline_shape_24613_spectra_normalised_gradients_vs_float64_oracle(dev, tmp_path):
    """c3's shape (4000 px, N_h = 16, masks), 24 576 + 37 spectra drawn from the model -- past the 96-spectra-per-CU rule at
    which the library switches pass 2 to its pixel-resident form BY ITSELF (flags = 0, qfa_host.h pass2_use_pixres), with a
    ragged last group: the NORMALISED gradients of one HIP launch against the float64 oracle summed over the same spectra
    on the host cores.  The forced forms (two-role k_grads_x, its three-product variant, k_grads_t) beside it."""
    import torch
    from qfa_amd import synthetic
    from tools import oracle_pool
    from tools import parity_sections as PS
    npix, nh, B = 4000, 16, 24576 + 37
    assert B >= 96 * torch.cuda.get_device_properties(dev).multi_processor_count      # the automatic dispatch is what runs
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    batch = PS.make_config_batch(p, mu, wav, nb, B, 20220733, dev, True)
    host = {k: x.cpu().numpy() for k, x in zip(("delta", "error", "zabs", "mask"), batch)}
    ol, og, sums, counts = oracle_pool.oracle_sums(p, host, str(tmp_path / "oracle"))
    out = {}
    for name, fl in (("default", 0), ("six", _lib.F_PASS2_XDL), ("pixres", _lib.F_PASS2_PIXRES)):
        m = _model(dev, p, mu, nb, nr, nh, fl)
        acc = m.accumulate(*batch).clone()
        loss, g = m._finalize(acc, True)
        e = {k: rel_l2(g[k].cpu().numpy(), og[k]) for k in KEYS}
        e["loss"] = abs(loss.item() - ol) / abs(ol)
        d = g["F"].cpu().numpy().astype(np.float64) - og["F"]
        e["F_over_terms"] = np.linalg.norm(d) / _gF_terms_scale(m, acc)
        e["cancellation"] = _gF_terms_scale(m, acc) / np.linalg.norm(og["F"])
        out[name] = e
    print("24613 x 4000, N_h = 16 vs float64 oracle:", out)
    six = out["six"]
    assert six["loss"] < 6e-7
    # achieved (round 5: fresh accumulators per tile in pass 1, float16 pieces in stages 1 and 3 of pass 2): k_grads_t F 1.61e-5
    # (6.8e-7 of the cancelling sums, which are 24x the gradient), k_grads_x 1.78e-5 / 7.5e-7; Psi 3.1e-6, omega 6.7e-6, loss 1.8e-7.
    # (3.2e-5 with pass 1's chains of rounds 1-4 -- and again with pass 1 on float16 pieces, QFA_P1_F16, which is why that
    # option is off; round 2's three-product form was at 5.5e-5.)  The bars sit 40-60 % above what is achieved: the arithmetic is
    # deterministic at this size (no atomics), so a change of these numbers is a change of the code.
    assert six["F"] < 2.5e-5 and six["F_over_terms"] < 1.1e-6, six
    assert six["Psi"] < 8e-6 and six["omega"] < 1.2e-5, six
    # the pixel-resident form (k_grads_t: W accumulated in float32 MFMA registers over 2 500 spectra per range, F applied once
    # at the end) meets the same bars
    for name in ("pixres", "default"):
        t = out[name]
        assert t["loss"] < 6e-7 and t["F"] < 2.5e-5 and t["F_over_terms"] < 1.1e-6 and t["Psi"] < 8e-6 and t["omega"] < 1.2e-5, (name, t)
    # flags = 0 took the pixel-resident form: bit for bit the forced launch (its sums leave through slab rows in fixed order)
    assert out["default"] == out["pixres"]
    del batch
    torch.cuda.empty_cache()


def test_c5_shape_4096_spectra_normalised_gradients_vs_float64_oracle(dev, tmp_path):
    """BASELINE configs[4]'s shape (8000 px, N_h = 32, masks): the NORMALISED gradients of one HIP launch over 4 096 spectra
    against the float64 oracle summed over the same spectra on the host cores (the oracle runs ~60 spectra/s there: the
    20 000-spectrum run of the same comparison is tools/c5_probe.py, its output profiles/r4_c5_precision.txt: F 8.3e-5).
    From round 4 on stage 3 at N_h = 17..32 issues six piece products too (k_grads_s3<32, 6>) and pass 1 cuts its
    accumulation chains at 63 tiles (qfa_host.h, QFA_P1_MAX_CHAIN: 3.6e-4 -> 8.3e-5 on the 20 000 spectra)."""
    import torch
    from qfa_amd import synthetic
    from tools import oracle_pool
    from tools import parity_sections as PS
    npix, nh, B = 8000, 32, 4096
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    batch = PS.make_config_batch(p, mu, wav, nb, B, 20220755, dev, True)
    host = {k: x.cpu().numpy() for k, x in zip(("delta", "error", "zabs", "mask"), batch)}
    ol, og, sums, counts = oracle_pool.oracle_sums(p, host, str(tmp_path / "oracle"))
    del host
    out = {}
    for name, fl in (("six", 0), ("fast", _lib.F_S3_FAST)):
        m = _model(dev, p, mu, nb, nr, nh, fl)
        acc = m.accumulate(*batch).clone()
        loss, g = m._finalize(acc, True)
        e = {k: rel_l2(g[k].cpu().numpy(), og[k]) for k in KEYS}
        e["loss"] = abs(loss.item() - ol) / abs(ol)
        d = g["F"].cpu().numpy().astype(np.float64) - og["F"]
        e["F_over_terms"] = np.linalg.norm(d) / _gF_terms_scale(m, acc)
        e["cancellation"] = _gF_terms_scale(m, acc) / np.linalg.norm(og["F"])
        out[name] = e
    print("4096 x 8000, N_h = 32 vs float64 oracle:", out)
    six, fast = out["six"], out["fast"]
    assert six["loss"] < 2e-6
    # achieved: F 4.5e-5 here (6.4e-6 of the cancelling sums, which are 7x the gradient at 4 096 spectra) with pass 1's chains cut
    # at 32 tiles (round 5; 8.2e-5 at 63 tiles, 3.6e-4 uncut): the bar VERDICT r4 item 2 asked for
    assert six["F"] < 8e-5 and six["F_over_terms"] < 1.2e-5, six
    assert six["Psi"] < 2e-5 and six["omega"] < 2e-5, six
    assert six["F"] <= 1.05 * fast["F"] + 1e-6, (six["F"], fast["F"])      # six piece products are never the worse form
    del batch
    torch.cuda.empty_cache()
