"""Device-side batch builder (next row N1) against the golden outputs of the reference's host
preprocessing (tau_total over one and two Lyman series, zabs, mu estimate + boxcar, delta) and
against the oracle; then QFA.train driven by it."""
import numpy as np
import pytest

from conftest import golden, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["c1", "lyb"])
def test_batch_builder_matches_reference_preprocessing(tag):
    import torch
    from qfa_amd.dataloader import DeviceDataloader
    dev = torch.device("cuda:0")
    g = golden("g11_dataprep.npz")
    wav, flux, zq = g[f"wav_{tag}"], g[f"flux_{tag}"], g["zqso"]
    err = np.where(flux != -999.0, 0.1, -999.0)
    dl = DeviceDataloader(flux, err, zq, wav, batch_size=4, device=dev, shuffle=False)
    nb = int(np.sum(wav < 1215.67))
    assert (dl.Nb, dl.Nr, dl.data_size, len(dl)) == (nb, len(wav) - nb, 6, 6)
    # flux is held in float32 on the device: compare against float32-rounded references
    assert rel_l2(dl._mu_raw, g[f"mu_raw_{tag}"]) < 2e-7
    assert rel_l2(dl.mu, g[f"mu_{tag}"]) < 2e-7
    dl.rewind()
    rows = []
    while dl.have_next_batch():
        rows.append([t.cpu().numpy() for t in dl.next_batch()])
    assert [r[0].shape[0] for r in rows] == [4, 2]
    delta = np.concatenate([r[0] for r in rows])
    e_out = np.concatenate([r[1] for r in rows])
    zabs = np.concatenate([r[2] for r in rows])
    mask = np.concatenate([r[3] for r in rows])
    assert delta.dtype == np.float32 and zabs.dtype == np.float32 and mask.dtype == bool
    assert np.array_equal(zabs, g[f"zabs_{tag}"].astype(np.float32))
    assert np.array_equal(mask, (flux != -999.0))
    assert np.array_equal(e_out, err.astype(np.float32))
    ok = mask
    assert np.max(np.abs(delta[ok] - g[f"delta_{tag}"][ok])) < 5e-7 * np.max(np.abs(g[f"delta_{tag}"][ok]))
    # __getitem__: raw flux + the same zabs / mask (predict loop of main.py:94)
    f0, e0, z0, m0, p0 = dl[3]
    assert np.array_equal(f0.cpu().numpy(), flux[3].astype(np.float32))
    assert np.array_equal(z0.cpu().numpy(), zabs[3]) and np.array_equal(m0.cpu().numpy(), mask[3])


def test_train_with_device_dataloader_matches_host_prepared_batches(tmp_path):
    import torch
    from oracle import qfa_oracle as O
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    from qfa_amd.dataloader import DeviceDataloader
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(320)
    p, mu0 = synthetic.mock_parameters(320, nb, 4, seed=5)
    b = synthetic.make_batch_numpy(p, mu0, wav, nb, 12, seed=51, masks=False)
    dl = DeviceDataloader(b["flux"], b["error"], b["zqso"], wav, batch_size=6, device=dev, shuffle=False)
    raw, mu = O.mu_estimate(wav, b["flux"].astype(np.float64), b["flux"] != -999.0, b["zqso"], nb)
    assert rel_l2(dl.mu, mu) < 1e-6
    model = QFA(nb, nr, 4, dev, model_params=p)
    opt = Adam(model.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)
    model.train(opt, dl, 1, str(tmp_path), quiet=True)
    # replica with host-prepared batches through the oracle
    delta = O.delta_from_flux(wav, b["flux"].astype(np.float64), b["zqso"], mu, nb).astype(np.float32)
    zabs = O.zabs_from_zqso(wav, b["zqso"], nb).astype(np.float32)
    params = {k: np.asarray(v, dtype=np.float64) for k, v in p.items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v = {k: np.zeros_like(vv) for k, vv in params.items()}
    for s in (0, 6):
        _, g = O.forward(params, delta[s:s + 6], b["error"][s:s + 6], zabs[s:s + 6], b["mask"][s:s + 6])
        params, m, v = O.adam_update(m, v, 0, params, g, O.step_lr(0, 1e-3, 0.9, 10), weight_decay=1e-1)
        params = O.clip_params(params)
    for k in params:
        assert rel_l2(model.parameters[k].cpu().numpy(), params[k]) < 2e-5, k
    assert rel_l2(model.mu.cpu().numpy(), mu) < 1e-6


def test_reference_loader_methods_get_rows_sample_set_device_set_tau():
    """the rest of the reference Dataloader surface main.py touches (QFA/dataloader.py:140-179): set_device (main.py:63),
    set_tau, sample; plus get_rows (one launch per slice for the batched predict writer)"""
    import torch
    from functools import partial
    from qfa_amd import utils
    from qfa_amd.dataloader import DeviceDataloader
    dev = torch.device("cuda:0")
    g = golden("g11_dataprep.npz")
    wav, flux, zq = g["wav_c1"], g["flux_c1"], g["zqso"]
    err = np.where(flux != -999.0, 0.1, -999.0)
    dl = DeviceDataloader(flux, err, zq, wav, batch_size=4, device=dev, shuffle=False)
    dl.set_device(dev)
    dl.set_device(torch.device("cuda"))
    with pytest.raises(Exception):
        dl.set_device(torch.device("cuda:7"))
    f, e, z, m, paths = dl.get_rows(1, 5)
    assert f.shape[0] == 4 and list(paths) == [1, 2, 3, 4]
    for r, i in enumerate(range(1, 5)):
        fi, ei, zi, mi, pi = dl[i]
        assert torch.equal(f[r], fi) and torch.equal(e[r], ei) and torch.equal(z[r], zi) and torch.equal(m[r], mi)
    np.random.seed(3)
    d, e2, z2, m2 = dl.sample()
    assert d.shape == (4, len(wav)) and m2.dtype == torch.bool
    np.random.seed(3)
    sig = np.random.randint(0, 6, size=(4,))
    dref = dl._build(sig)[0]
    assert torch.equal(d, dref)
    mu_becker = dl.mu.copy()
    dl.set_tau(partial(utils.tau, which="fg"))
    assert not np.allclose(dl.mu, mu_becker)
    dl.set_tau("becker")
    assert np.array_equal(dl.mu, mu_becker)
