"""The resident, indexed input form of a batch (include/qfa_hip.h, ABI v3: ``qfa_batch_t.rows`` / ``row_stride``).

The reference materialises every batch on the host (reference QFA/dataloader.py:124-138) and shuffles an epoch by
permuting the whole data set (:154-167).  ``ResidentBatch`` names B spectra of a data set that stays where it is in
HBM instead: the arrays of the WHOLE set -- ``delta = flux - mu * exp(-tau_total)`` and the bool mask, built once per
loader, ``flux`` and ``error`` as uploaded, rows ``stride`` elements apart -- plus the device array ``rows`` of the B row
numbers.  ``QFA.forward / step / predict(batch=...)`` hand exactly these pointers to the kernels: nothing is gathered,
copied or rebuilt per step.
"""
from __future__ import annotations

import torch


class ResidentBatch(object):
    __slots__ = ("flux", "delta", "error", "mask", "zq1", "pix_ratio", "rows", "stride", "Npix", "Nb", "zabs")

    def __init__(self, flux, delta, error, mask, zq1, pix_ratio, rows, Npix, Nb, zabs=None):
        """flux, delta, error: float32 (N, stride) storage (contiguous; the first Npix pixels of a row are used; either of
        flux / delta may be None when only training / only prediction is meant); mask: bool (N, stride);
        zq1: float32 (N,) = 1 + z_qso and pix_ratio: float32 (Nb,) = wav_blue / 1215.67 (the factored-z form), or
        zabs: float32 (N, Nb) resident absorber redshifts; rows: int32 (B,) device tensor of row numbers (a contiguous
        slice of a permutation is fine)."""
        self.flux, self.delta, self.error, self.mask = flux, delta, error, mask
        self.zq1, self.pix_ratio, self.rows, self.zabs = zq1, pix_ratio, rows, zabs
        self.stride = int(error.shape[1])
        self.Npix, self.Nb = int(Npix), int(Nb)

    @property
    def B(self):
        return int(self.rows.shape[0])

    def with_rows(self, rows):
        """the same resident arrays, other rows"""
        return ResidentBatch(self.flux, self.delta, self.error, self.mask, self.zq1, self.pix_ratio, rows, self.Npix, self.Nb,
                             self.zabs)

    def materialize(self, raw_flux=False):
        """(delta | flux, error, zabs, mask), zfac of these rows as contiguous tensors in batch order: the 4-tuple of the
        reference's ``next_batch`` contract (reference QFA/dataloader.py:124-138) and the factors the kernels take in its
        place.  Pure indexing -- tests compare the indexed form with this gathered copy of it."""
        idx = self.rows.long()
        src = self.flux if raw_flux else self.delta
        d = src[idx, :self.Npix].contiguous()
        e = self.error[idx, :self.Npix].contiguous()
        m = self.mask[idx, :self.Npix].contiguous()
        z = self.zabs[idx].contiguous() if self.zabs is not None else None
        zfac = (self.zq1[idx].contiguous(), self.pix_ratio) if self.zq1 is not None else None
        return (d, e, z, m), zfac
