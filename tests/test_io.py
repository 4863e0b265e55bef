"""On-disk spectrum reader + catalogue selection (next row N3): host logic on CPU, then through the device
dataloader on the GPU.  The selection is compared with a restatement of reference QFA/dataloader.py:48-55 under
the same numpy seed; the reader with the reference's own shipped spectrum file (tests/golden/sdss_spectrum.npz)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN as GOLDEN_DIR, golden


def _write_spectra(d, n, npix, seed=3):
    rng = np.random.default_rng(seed)
    names = []
    for i in range(n):
        flux = rng.normal(1.0, 0.2, npix)
        err = np.full(npix, 0.1)
        flux[rng.random(npix) < 0.05] = -999.0
        err[rng.random(npix) < 0.02] = -999.0
        name = f"spec-{i:03d}.npz"
        np.savez(os.path.join(d, name), flux=flux, error=err, z=2.2 + 0.1 * i)
        names.append(name)
    return names


def test_read_spectra_order_dtypes_and_grid_check(tmp_path):
    from qfa_amd import io
    names = _write_spectra(tmp_path, 5, 64)
    paths = [os.path.join(tmp_path, n) for n in names]
    f1, e1, z1, p1 = io.read_spectra(paths, nprocs=1)
    f4, e4, z4, p4 = io.read_spectra(paths, nprocs=4)
    assert f1.dtype == np.float32 and e1.dtype == np.float32 and z1.dtype == np.float64
    assert f1.shape == (5, 64) and np.array_equal(f1, f4) and np.array_equal(e1, e4) and np.array_equal(z1, z4)
    assert list(p1) == paths and list(p4) == paths
    assert np.allclose(z1, 2.2 + 0.1 * np.arange(5))
    raw = np.load(paths[2])
    assert np.array_equal(f1[2], raw["flux"].astype(np.float32))
    assert np.array_equal(f1[2] == -999.0, raw["flux"] == -999.0)          # the sentinel survives float32
    np.savez(os.path.join(tmp_path, "short.npz"), flux=np.ones(32), error=np.ones(32), z=2.0)
    with pytest.raises(ValueError):
        io.read_spectra(paths + [os.path.join(tmp_path, "short.npz")])
    with pytest.raises(ValueError):
        io.read_spectra([])


def test_reads_the_reference_spectrum_file():
    from qfa_amd import io
    path = os.path.join(GOLDEN_DIR, "sdss_spectrum.npz")
    flux, error, z, p = io.read_spectra([path])
    g = golden("sdss_spectrum.npz")
    assert flux.shape == (1, 1913) and np.array_equal(flux[0], g["flux"].astype(np.float32))
    assert np.array_equal(error[0], g["error"].astype(np.float32)) and z[0] == float(g["z"])
    wav = io.wavelength_grid(1030.0, 1600.0, 1e-4)
    assert len(wav) == 1913                                                # the grid of the shipped model


def test_catalog_selection_matches_reference_semantics(tmp_path):
    import pandas as pd
    from qfa_amd import io
    rng = np.random.default_rng(0)
    n = 40
    cat = pd.DataFrame({"file": [f"s{i}.npz" for i in range(n)], "snr": rng.uniform(0, 10, n),
                        "z": rng.uniform(2.0, 3.5, n), "num_mask": rng.integers(0, 200, n)})
    cpath = os.path.join(tmp_path, "catalog.csv")
    cat.to_csv(cpath, index=False)
    lim = dict(snr_min=2.0, snr_max=9.0, z_min=2.1, z_max=3.2, num_mask=150)

    def reference_pick(num):                # QFA/dataloader.py:48-51
        c = pd.read_csv(cpath)
        crit = ((c["snr"] >= lim["snr_min"]) & (c["snr"] <= lim["snr_max"]) & (c["z"] >= lim["z_min"])
                & (c["z"] <= lim["z_max"]) & (c["num_mask"] <= lim["num_mask"]))
        return np.random.choice(c["file"][crit].values, size=(num,), replace=(np.sum(crit) < num)), int(np.sum(crit))

    for num in (5, 200):                     # without and with replacement
        np.random.seed(11)
        want, nok = reference_pick(num)
        np.random.seed(11)
        got = io.select_from_catalog(cpath, num, output_dir=os.path.join(tmp_path, "out"), prefix="train", **lim)
        assert list(got) == list(want)
        assert (len(set(got)) == num) == (nok >= num)
        written = pd.read_csv(os.path.join(tmp_path, "out", "train-catalog.csv"), header=None).values.squeeze()
        assert list(np.atleast_1d(written)) == list(want)
    assert io.read_prediction_catalog(os.path.join(tmp_path, "out", "train-catalog.csv"))[0] in set(cat["file"])
    with pytest.raises(ValueError):
        io.select_from_catalog(cpath, 3, 100.0, 200.0, 2.0, 3.0, 10)


@pytest.mark.gpu
def test_device_dataloader_from_files_and_catalog(tmp_path):
    import pandas as pd
    import torch
    from oracle import qfa_oracle as O
    from qfa_amd import io
    from qfa_amd.dataloader import DeviceDataloader
    dev = torch.device("cuda:0")
    wav = io.wavelength_grid(1030.0, 1600.0, 2e-3)
    npix, nb = len(wav), int(np.sum(wav < 1215.67))
    names = _write_spectra(tmp_path, 6, npix)
    paths = [os.path.join(tmp_path, n) for n in names]
    dl = DeviceDataloader.from_files(paths, wav, batch_size=4, device=dev, nprocs=2, shuffle=False)
    flux, err, zq, _ = io.read_spectra(paths)
    assert dl.data_size == 6 and list(dl.pathlist) == paths
    _, mu = O.mu_estimate(wav, flux.astype(np.float64), (flux != -999.0) & (err != -999.0), zq, nb)
    assert np.max(np.abs(dl.mu - mu)) < 1e-6 * np.max(np.abs(mu))
    delta, e, zabs, mask = [t.cpu().numpy() for t in dl.next_batch()]
    assert np.array_equal(mask, (flux[:4] != -999.0) & (err[:4] != -999.0))
    assert np.array_equal(zabs, O.zabs_from_zqso(wav, zq[:4], nb).astype(np.float32))
    want = O.delta_from_flux(wav, flux[:4].astype(np.float64), zq[:4], dl.mu, nb)
    assert np.max(np.abs(delta[mask] - want[mask])) < 2e-6 * np.max(np.abs(want[mask]))
    cat = pd.DataFrame({"file": names, "snr": 5.0, "z": zq, "num_mask": 0})
    cat.to_csv(os.path.join(tmp_path, "cat.csv"), index=False)
    np.random.seed(2)
    dl2 = DeviceDataloader.from_catalog(os.path.join(tmp_path, "cat.csv"), str(tmp_path), 4, wav, 4, dev, z_max=2.45,
                                        output_dir=os.path.join(tmp_path, "o"))
    assert dl2.data_size == 4 and all(os.path.basename(p) in names[:3] for p in dl2.pathlist)   # z <= 2.45: 3 files
