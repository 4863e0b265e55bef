"""Pin the CPU oracle: reference known-answer vectors + fixtures generated from the imported
reference (tests/golden/make_golden.py).  No GPU, no HIP library."""
import numpy as np
import pytest

from conftest import golden, rel_l2
from oracle import qfa_oracle as O
from qfa_amd import synthetic

KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")


def _predict_inputs(grid):
    wav, nb, nr = grid
    sp = golden("sdss_spectrum.npz")
    flux, error, z = sp["flux"], sp["error"], float(sp["z"])
    mask = (flux != -999.) & (error != -999.)
    zabs = wav[:nb] * (1 + z) / 1215.67 - 1
    return sp, flux, error, zabs, mask


def test_g1_shipped_known_answers_full_mask(shipped, grid):
    """Reference's own stored ll/h/our (data/spec-4321-55504-0114.npz), full mask (pin G-A)."""
    p, mu = shipped
    sp, flux, error, zabs, mask = _predict_inputs(grid)
    ll, hm, hc, cont, unc = O.predict_single(p, mu, flux, error, zabs.astype(np.float32), mask)
    assert abs(ll - float(sp["ll"])) / abs(float(sp["ll"])) < 2e-6
    assert rel_l2(hm, sp["h"].squeeze()) < 2e-5
    assert np.max(np.abs(cont - sp["our"]) / np.abs(sp["our"])) < 2e-6
    # stored variance = (A * unc)^2 (pin G-C)
    A, _, _ = O.pixel_terms(p, error, zabs.astype(np.float32))
    assert rel_l2((A * unc) ** 2, sp["our_uncertainty"]) < 1e-5


def test_g2_shipped_known_answers_red_only(shipped, grid):
    p, mu = shipped
    wav, nb, nr = grid
    sp, flux, error, zabs, mask = _predict_inputs(grid)
    mk = mask & (np.arange(len(wav)) >= nb)
    ll, hm, hc, cont, unc = O.predict_single(p, mu, flux, error, zabs.astype(np.float32), mk)
    assert abs(ll - float(sp["ll_red"])) / abs(float(sp["ll_red"])) < 2e-6
    assert rel_l2(hm, sp["h_red"].squeeze()) < 2e-5
    assert np.max(np.abs(cont - sp["our_red"]) / np.abs(sp["our_red"])) < 2e-6


def test_g1_g2_reference_import_outputs(shipped, grid):
    p, mu = shipped
    g = golden("g1_g2_predict.npz")
    sp, flux, error, _, _ = _predict_inputs(grid)
    for tag in ("full", "red"):
        ll, hm, hc, cont, unc = O.predict_single(p, mu, flux, error, g["zabs"], g[f"mask_{tag}"])
        assert abs(ll - float(g[f"ll_{tag}"].squeeze())) / abs(float(g[f"ll_{tag}"].squeeze())) < 2e-6
        assert rel_l2(hm, g[f"hmean_{tag}"].squeeze()) < 2e-5
        assert rel_l2(hc, g[f"hcov_{tag}"]) < 2e-5
        assert rel_l2(cont, g[f"cont_{tag}"]) < 1e-6
        assert rel_l2(unc, g[f"unc_{tag}"]) < 5e-6


def test_g3_single_spectrum_nll_and_grads(shipped, grid):
    p, mu = shipped
    wav, nb, nr = grid
    g = golden("g3_single.npz")
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 4, seed=int(g["seed"]))
    for s in range(4):
        nll, gr = O.nll_and_grads_single(p, b["delta"][s], b["error"][s], b["zabs"][s], b["mask"][s])
        assert abs(nll - g["nll"][s]) / abs(g["nll"][s]) < 2e-6
        assert rel_l2(gr["F"], g["g_F"][s]) < 1e-4
        assert rel_l2(gr["Psi"], g["g_Psi"][s]) < 5e-6
        assert rel_l2(gr["omega"], g["g_omega"][s]) < 5e-6
        for k in ("tau0", "c0", "beta"):
            assert abs(gr[k] - g[f"g_{k}"][s]) / abs(g[f"g_{k}"][s]) < 2e-5
        assert np.all(gr["F"][~b["mask"][s]] == 0)


def test_g4_forward_count_normalisation_nan_and_red_only(shipped, grid):
    p, mu = shipped
    wav, nb, nr = grid
    g = golden("g4_forward.npz")
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 8, seed=int(g["seed"]), red_only=(3,), dead_range=(900, 910))
    loss, gr = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    assert abs(loss - float(g["loss"].squeeze())) / abs(float(g["loss"].squeeze())) < 2e-6
    for k in KEYS:
        ref = g[f"g_{k}"]
        assert np.array_equal(np.isnan(gr[k]), np.isnan(ref)), k
        ok = ~np.isnan(ref)
        assert rel_l2(np.asarray(gr[k])[ok], ref[ok]) < (1e-4 if k == "F" else 2e-5), k
    assert np.isnan(g["g_F"][900:910]).all() and np.isnan(g["g_Psi"][900:910]).all()


def test_g5_full_step_adam_clip(shipped, grid):
    p, mu = shipped
    wav, nb, nr = grid
    g = golden("g5_step.npz")
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 128, seed=int(g["seed"]))
    loss, gr = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    assert abs(loss - float(g["loss"].squeeze())) / abs(float(g["loss"].squeeze())) < 2e-6
    for k in KEYS:
        assert rel_l2(gr[k], g[f"g_{k}"]) < (1e-4 if k == "F" else 2e-5), k
    zeros = {k: np.zeros_like(np.asarray(p[k], dtype=np.float64)) for k in KEYS}
    lr = O.step_lr(0, 1e-3, 0.9, 10)
    newp, m, v = O.adam_update(zeros, zeros, 0, p, gr, lr, weight_decay=1e-1)
    newp = O.clip_params(newp)
    for k in KEYS:
        assert rel_l2(newp[k], g[f"p_{k}"]) < 2e-6, k


def test_g6_smooth_and_clip(shipped):
    p, mu = shipped
    g = golden("g6_smooth_clip.npz")
    sm = O.smooth_params({k: np.asarray(v, dtype=np.float64) for k, v in p.items()})
    for k in ("F", "Psi", "omega"):
        assert rel_l2(sm[k], g[f"smooth_{k}"]) < 2e-6, k
    pre = {k: g[f"preclip_{k}"] for k in KEYS}
    cl = O.clip_params(pre)
    for k in KEYS:
        assert np.array_equal(np.asarray(cl[k], dtype=np.float32), g[f"clip_{k}"]), k


def test_g7_adam_trace_bias_correction_per_epoch():
    g = golden("g7_adam.npz")
    params = {k: g[f"init_{k}"].astype(np.float64) for k in KEYS}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v = {k: np.zeros_like(vv) for k, vv in params.items()}
    it, i = 0, 0
    for epoch in range(3):
        for _ in range(2):
            grads = {k: g[f"grad{it}_{k}"] for k in KEYS}
            params, m, v = O.adam_update(m, v, i, params, grads, O.step_lr(i, 1e-2, 0.9, 2), weight_decay=1e-3)
            for k in KEYS:
                assert rel_l2(params[k], g[f"p{it}_{k}"]) < 5e-6, (it, k)
            it += 1
        i += 1


def test_g8_woodbury_inverse_logdet():
    g = golden("g8_woodbury.npz")
    M, D = g["M"].astype(np.float64), g["D"].astype(np.float64)
    assert rel_l2(O.woodbury_inverse(M, D), g["inv"]) < 5e-6
    assert abs(O.woodbury_logdet(M, D) - float(g["logdet"])) / abs(float(g["logdet"])) < 2e-6
    dense = M @ M.T + np.diag(D)
    assert rel_l2(O.woodbury_inverse(M, D), np.linalg.inv(dense)) < 1e-10
    assert abs(O.woodbury_logdet(M, D) - np.linalg.slogdet(dense)[1]) < 1e-9


def test_g9_tau_functions():
    g = golden("g9_tau.npz")
    z = g["z"].astype(np.float64)
    for which in ("becker", "fg", "kamble", "mock"):
        for series in (1, 2, 5):
            assert rel_l2(O.tau_eff(z, which, series), g[f"tau_{which}_{series}"]) < 2e-6
    assert rel_l2(O.tau_hi(z, np.float32(0.0123), np.float32(3.1)), g["tauHI"]) < 2e-6
    assert rel_l2(O.omega_zdep(z, np.float32(0.0123), np.float32(3.1), np.float32(0.27)), g["omega_func"]) < 5e-6
    with pytest.raises(NotImplementedError):
        O.tau_eff(z, "nope")


def test_g10_k16_reference_overflows_oracle_finite():
    g = golden("g10_k16.npz")
    assert np.isinf(g["loss"]).all()           # quirk Q7: float32 det overflow in the reference
    n_pix = int(g["n_pix"])
    wav, nb, nr = synthetic.wavelength_grid(n_pix)
    r16 = np.random.default_rng(16)
    p16 = {"F": (r16.random((n_pix, 16)) - 0.5).astype(np.float32), "Psi": np.ones(n_pix, np.float32),
           "omega": np.ones(nb, np.float32), "tau0": np.float32(0.02), "c0": np.float32(0.3),
           "beta": np.float32(2.0)}
    _, mu16 = synthetic.mock_parameters(n_pix, nb, 16, seed=16)
    b = synthetic.make_batch_numpy(p16, mu16, wav, nb, 2, seed=int(g["seed"]))
    loss, gr = O.forward(p16, b["delta"], b["error"], b["zabs"], b["mask"])
    assert np.isfinite(loss)
    for k in KEYS:
        ref = g[f"g_{k}"]
        ok = ~np.isnan(ref)
        assert rel_l2(np.asarray(gr[k])[ok], ref[ok]) < (3e-4 if k == "F" else 5e-5), k


def test_dense_port_matches_lowrank_oracle(shipped, grid):
    """The dense O(n^3) CPU port (bench cpu_baseline) agrees with the low-rank oracle."""
    import torch
    from oracle import dense_port as DP
    p, mu = shipped
    wav, nb, nr = grid
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 3, seed=99, red_only=(1,))
    P = DP.to_torch_params(p)
    loss_d, g_d = DP.dense_forward(P, torch.tensor(b["delta"]), torch.tensor(b["error"]),
                                   torch.tensor(b["zabs"]), torch.tensor(b["mask"]))
    loss, gr = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    assert abs(float(loss_d) - loss) / abs(loss) < 5e-6
    for k in KEYS:
        ref = np.asarray(gr[k])
        ok = ~np.isnan(ref)
        assert rel_l2(g_d[k].numpy()[ok], ref[ok]) < (2e-4 if k == "F" else 5e-5), k


def test_oracle_float32_mode_close_to_float64(shipped, grid):
    p, mu = shipped
    wav, nb, nr = grid
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 2, seed=5)
    n64, g64 = O.nll_and_grads_single(p, b["delta"][0], b["error"][0], b["zabs"][0], b["mask"][0])
    n32, g32 = O.nll_and_grads_single(p, b["delta"][0], b["error"][0], b["zabs"][0], b["mask"][0], dtype=np.float32)
    assert abs(n32 - n64) / abs(n64) < 1e-4
    assert rel_l2(g32["Psi"], g64["Psi"]) < 1e-3


def test_g11_dataloader_preprocessing():
    g = golden("g11_dataprep.npz")
    zq = g["zqso"]
    for tag in ("c1", "lyb"):
        wav = g[f"wav_{tag}"]
        nb = int(np.sum(wav < 1215.67))
        for which in ("becker", "kamble"):
            tt = O.tau_total(wav, zq, which)
            assert tt.shape == g[f"tau_total_{tag}_{which}"].shape
            assert rel_l2(tt, g[f"tau_total_{tag}_{which}"]) < 1e-12
        assert rel_l2(O.zabs_from_zqso(wav, zq, nb), g[f"zabs_{tag}"]) < 1e-14
        flux = g[f"flux_{tag}"]
        raw, mu = O.mu_estimate(wav, flux, flux != -999.0, zq, nb)
        assert rel_l2(raw, g[f"mu_raw_{tag}"]) < 1e-12
        assert rel_l2(mu, g[f"mu_{tag}"]) < 1e-12
        assert rel_l2(O.delta_from_flux(wav, flux, zq, mu, nb), g[f"delta_{tag}"]) < 1e-12
    # the second grid reaches below Ly-beta: two series contribute there
    assert int(np.sum(g["wav_lyb"][0] < O._LYMAN_LAM)) == 2


def test_g13_desi_model_reference_import_outputs():
    """The reference's second shipped model (data/model_parameters_desi.npz: N_pix = 9243, N_b = 2238, N_h = 8): one
    prediction (full mask, blue side masked) and one single-spectrum NLL + gradients from the imported reference
    (tests/golden/make_golden_desi.py) -- the one real large-N_pix workload the reference ships."""
    import os
    from conftest import GOLDEN
    p, mu = O.load_params_npz(os.path.join(GOLDEN, "model_parameters_desi.npz"))
    wav, nb, nr = synthetic.desi_grid()
    assert (len(wav), nb, p["F"].shape) == (9243, 2238, (9243, 8))
    g = golden("g13_desi.npz")
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 2, seed=int(g["seed"]))
    for tag, mk in (("full", b["mask"][0]), ("red", b["mask"][0] & (np.arange(len(wav)) >= nb))):
        ll, hm, hc, cont, unc = O.predict_single(p, mu, b["flux"][0], b["error"][0], b["zabs"][0], mk)
        assert abs(ll - float(g[f"ll_{tag}"].squeeze())) / abs(float(g[f"ll_{tag}"].squeeze())) < 5e-6
        assert rel_l2(hm, g[f"hmean_{tag}"].squeeze()) < 1e-4
        assert rel_l2(hc, g[f"hcov_{tag}"]) < 1e-4
        assert rel_l2(cont, g[f"cont_{tag}"]) < 2e-6
        assert rel_l2(unc, g[f"unc_{tag}"]) < 2e-5
    nll, gr = O.nll_and_grads_single(p, b["delta"][1], b["error"][1], b["zabs"][1], b["mask"][1])
    assert abs(nll - float(g["nll"].squeeze())) / abs(float(g["nll"].squeeze())) < 5e-6
    assert rel_l2(gr["F"], g["g_F"]) < 3e-4
    assert rel_l2(gr["Psi"], g["g_Psi"]) < 2e-5 and rel_l2(gr["omega"], g["g_omega"]) < 2e-5
    for k in ("tau0", "c0", "beta"):
        assert abs(gr[k] - g[f"g_{k}"]) / abs(g[f"g_{k}"]) < 2e-4, k
