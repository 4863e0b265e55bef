"""Device-side dataloader.

``DeviceDataloader`` offers the contract ``QFA.train`` consumes from the reference's
``Dataloader`` (reference QFA/dataloader.py:114-138,154-167,189-191: ``.mu``, ``.data_size``,
``.batch_size``, ``.rewind()``, ``.have_next_batch()``, ``.next_batch()``, ``len()``,
``__getitem__``) for spectra that are already in memory: flux, error and redshifts stay resident
in HBM, and what the reference computes on the host with numpy for every batch -- ``zabs``,
``tau_total`` over the Lyman series, ``delta = flux - mu * exp(-tau_total)``, the bool mask, and
the smoothed mean continuum ``mu`` -- runs in HIP kernels (``qfa_build_batch_f32``,
``qfa_mu_estimate_f64``).  Reading spectra from disk and the catalogue selection (row N3) are host code in ``qfa_amd.io``
(``from_files`` / ``from_catalog`` below).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

f32 = torch.float32
LYA = 1215.67


class DeviceDataloader(object):

    def __init__(self, flux, error, zqso, wav_grid, batch_size, device, tau="becker", window_length_for_mu=16,
                 shuffle=True, mode="train", paths=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.QFAHipError("DeviceDataloader needs a HIP device (torch device 'cuda')")
        if tau not in _lib.TAU_IDS:
            raise NotImplementedError("currently available mean optical depth function: ['becker', 'fg', 'kamble']")
        self._which = _lib.TAU_IDS[tau]
        self.wav_grid = np.asarray(wav_grid, dtype=np.float64)
        self.Nb = int(np.sum(self.wav_grid < LYA))                      # dataloader.py:62
        self.Nr = len(self.wav_grid) - self.Nb
        self.Npix = len(self.wav_grid)
        self.type = mode
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.flux = torch.as_tensor(np.asarray(flux, dtype=np.float32), device=self.device).contiguous()
        self.error = torch.as_tensor(np.asarray(error, dtype=np.float32), device=self.device).contiguous()
        if tuple(self.flux.shape) != tuple(self.error.shape) or self.flux.shape[1] != self.Npix:
            raise _lib.QFAHipError("flux / error must be (N, len(wav_grid))")
        self.zqso = np.asarray(zqso, dtype=np.float64).reshape(-1)
        self.data_size = int(self.flux.shape[0])
        self.pathlist = np.asarray(paths) if paths is not None else np.arange(self.data_size)
        self._zq_dev = torch.as_tensor(self.zqso, device=self.device)
        self._wav_dev = torch.as_tensor(self.wav_grid, device=self.device)
        self._order = np.arange(self.data_size)
        self.cur = 0
        self._mu_raw, self._mu = self._estimate_mu(int(window_length_for_mu))
        self._mu_dev = torch.as_tensor(self._mu, device=self.device)

    # ------------------------------------------------------------------ from disk (row N3)
    @classmethod
    def from_files(cls, paths, wav_grid, batch_size, device, nprocs=1, **kw):
        """spectra read from per-spectrum .npz files (reference QFA/dataloader.py:18-45,84-88)"""
        from . import io
        flux, error, zqso, plist = io.read_spectra(paths, nprocs)
        return cls(flux, error, zqso, wav_grid, batch_size, device, paths=plist, **kw)

    @classmethod
    def from_catalog(cls, catalog, data_dir, num, wav_grid, batch_size, device, snr_min=-np.inf, snr_max=np.inf,
                     z_min=-np.inf, z_max=np.inf, num_mask=np.inf, nprocs=1, output_dir=None, prefix="train", **kw):
        """catalogue selection + read (reference QFA/dataloader.py:48-55,72-76)"""
        from . import io
        flux, error, zqso, plist = io.load_from_catalog(catalog, data_dir, num, snr_min, snr_max, z_min, z_max,
                                                        num_mask, nprocs, output_dir, prefix)
        return cls(flux, error, zqso, wav_grid, batch_size, device, paths=plist, **kw)

    # ------------------------------------------------------------------ mu
    def _estimate_mu(self, window):
        scratch = torch.empty(2 * self.Npix, dtype=torch.float64, device=self.device)
        raw = torch.empty(self.Npix, dtype=torch.float64, device=self.device)
        sm = torch.empty(self.Npix, dtype=torch.float64, device=self.device)
        _lib.check(_lib.lib().qfa_mu_estimate_f64(
            C.c_void_p(self.flux.data_ptr()), C.c_void_p(self.error.data_ptr()), C.c_void_p(self._zq_dev.data_ptr()),
            C.c_void_p(self._wav_dev.data_ptr()), float(self.wav_grid[0]), self._which, self.data_size, self.Npix,
            self.Nb, window, C.c_void_p(scratch.data_ptr()), C.c_void_p(raw.data_ptr()), C.c_void_p(sm.data_ptr()),
            _lib.current_stream(self.device)), "qfa_mu_estimate_f64")
        return raw.cpu().numpy(), sm.cpu().numpy()

    @property
    def mu(self):
        return self._mu

    # ------------------------------------------------------------------ batches
    def _build(self, rows, out=None):
        n = len(rows)
        idx = torch.as_tensor(np.asarray(rows, dtype=np.int32), device=self.device)
        if out is not None:                                   # caller-owned buffers (a captured step graph reads them)
            delta, err, zabs, mask = out
            if tuple(delta.shape) != (n, self.Npix) or tuple(zabs.shape) != (n, self.Nb):
                raise _lib.QFAHipError("out buffers do not match the batch")
        else:
            delta = torch.empty((n, self.Npix), dtype=f32, device=self.device)
            err = torch.empty((n, self.Npix), dtype=f32, device=self.device)
            zabs = torch.empty((n, self.Nb), dtype=f32, device=self.device)
            mask = torch.empty((n, self.Npix), dtype=torch.bool, device=self.device)
        _lib.check(_lib.lib().qfa_build_batch_f32(
            C.c_void_p(self.flux.data_ptr()), C.c_void_p(self.error.data_ptr()), C.c_void_p(self._zq_dev.data_ptr()),
            C.c_void_p(idx.data_ptr()), C.c_void_p(self._wav_dev.data_ptr()), float(self.wav_grid[0]),
            C.c_void_p(self._mu_dev.data_ptr()), self._which, n, self.Npix, self.Nb, C.c_void_p(delta.data_ptr()),
            C.c_void_p(err.data_ptr()), C.c_void_p(zabs.data_ptr()) if self.Nb > 0 else None,
            C.c_void_p(mask.data_ptr()), _lib.current_stream(self.device)), "qfa_build_batch_f32")
        return delta, err, zabs, mask

    def have_next_batch(self):
        return self.cur < self.data_size

    def next_batch(self, out=None):
        """delta, error, zabs, mask of the next batch (reference QFA/dataloader.py:124-138); ``out`` = four
        caller-owned tensors of the batch's shape to build into."""
        start = self.cur
        end = min(self.cur + self.batch_size, self.data_size)
        self.cur = end
        return self._build(self._order[start:end], out)

    def next_batch_size(self):
        return min(self.cur + self.batch_size, self.data_size) - self.cur

    def rewind(self):
        """shuffle and reset (reference QFA/dataloader.py:154-167); only the row order is permuted,
        the spectra stay where they are in HBM."""
        if self.shuffle:
            np.random.shuffle(self._order)
        self.cur = 0

    def __len__(self):
        return self.data_size

    def __getitem__(self, i):
        """raw flux (not delta), error, zabs, mask, path of spectrum i (reference dataloader.py:184-187)."""
        _, err, zabs, mask = self._build([int(i)])
        return self.flux[int(i)], err[0], zabs[0], mask[0], self.pathlist[int(i)]
