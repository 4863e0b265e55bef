"""On-disk spectrum reader and catalogue selection.

Host-side ingestion only: per-spectrum ``.npz`` files (``flux``, ``error``, ``z``; missing pixels are
``-999.``) and the catalogue filter of the reference (reference QFA/dataloader.py:18-55,72-90).  What the
reference does with the arrays afterwards (``zabs``, ``delta``, masks, ``mu``) runs on the device in
``qfa_amd.dataloader.DeviceDataloader``; this module only gets the bytes off the disk.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

MISSING = -999.0


def wavelength_grid(lam_min, lam_max, loglam_delta):
    """the common rest-frame grid (reference QFA/dataloader.py:59)"""
    return 10 ** np.arange(np.log10(lam_min), np.log10(lam_max), loglam_delta)


def read_npz_spectrum(path):
    """flux, error, z, path of one spectrum file (reference QFA/dataloader.py:18-30; the mask
    ``(flux != -999) & (error != -999)`` is derived on the device from the same sentinels)."""
    with np.load(path) as f:
        return np.asarray(f["flux"]), np.asarray(f["error"]), float(f["z"]), path


def read_spectra(paths, nprocs=1):
    """(flux (N, Npix) f32, error (N, Npix) f32, zqso (N,) f64, paths) in the order of ``paths``
    (reference QFA/dataloader.py:33-45; threads instead of a process pool: the work is file I/O and
    zip inflation, both release the GIL)."""
    paths = list(paths)
    if not paths:
        raise ValueError("no spectrum files given")
    if nprocs and nprocs > 1:
        with ThreadPoolExecutor(max_workers=int(nprocs)) as ex:
            data = list(ex.map(read_npz_spectrum, paths))
    else:
        data = [read_npz_spectrum(p) for p in paths]
    npix = {d[0].shape[-1] for d in data} | {d[1].shape[-1] for d in data}
    if len(npix) != 1:
        raise ValueError(f"all spectra must share one wavelength grid, got lengths {sorted(npix)}")
    flux = np.stack([d[0] for d in data]).astype(np.float32, copy=False)
    error = np.stack([d[1] for d in data]).astype(np.float32, copy=False)
    zqso = np.array([d[2] for d in data], dtype=np.float64)
    return flux, error, zqso, np.array([d[3] for d in data])


def select_from_catalog(catalog, num, snr_min, snr_max, z_min, z_max, num_mask, output_dir=None, prefix="train"):
    """File names drawn from a catalogue csv with columns file, snr, z, num_mask
    (reference QFA/dataloader.py:48-55): the rows inside the S/N, redshift and masked-pixel limits,
    ``num`` of them drawn with ``np.random.choice`` (with replacement only when fewer than ``num``
    qualify -- the global numpy RNG, as the reference), and the draw written to
    ``<output_dir>/<prefix>-catalog.csv``."""
    import pandas as pd
    cat = pd.read_csv(catalog)
    ok = ((cat["snr"] >= snr_min) & (cat["snr"] <= snr_max) & (cat["z"] >= z_min) & (cat["z"] <= z_max)
          & (cat["num_mask"] <= num_mask))
    pool = cat["file"][ok].values
    if len(pool) == 0:
        raise ValueError("no catalogue row passes the selection")
    files = np.random.choice(pool, size=(num,), replace=(int(np.sum(ok)) < num))
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        pd.Series(files).to_csv(os.path.join(output_dir, f"{prefix}-catalog.csv"), header=False, index=False)
    return files


def read_prediction_catalog(catalog):
    """the file list of ``--type predict`` (reference QFA/dataloader.py:86-87): every value of the csv"""
    import pandas as pd
    return list(np.atleast_1d(pd.read_csv(catalog).values.squeeze()))


def load_from_catalog(catalog, data_dir, num, snr_min, snr_max, z_min, z_max, num_mask, nprocs=1, output_dir=None,
                      prefix="train"):
    """select + read (reference QFA/dataloader.py:48-55)"""
    files = select_from_catalog(catalog, num, snr_min, snr_max, z_min, z_max, num_mask, output_dir, prefix)
    return read_spectra([os.path.join(data_dir, x) for x in files], nprocs)
