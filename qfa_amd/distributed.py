"""Data parallelism over spectra; no counterpart in the reference, which is
single-process (its loop over spectra is QFA/model.py:98-103).

Spectra are independent given the parameters, so every rank accumulates the raw sums of its own
shard into the packed buffer laid out by ``qfa_accum_floats`` (include/qfa_hip.h):

    [ accF (Npix*Nh) | sumA (Npix) | gPsi (Npix) | gOmega (Nb) | cnt (Npix) |
      g_tau0, g_c0, g_beta, n_spectra_with_blue, sum_nll, n_spectra, 0, 0 ]

and ONE all-reduce(sum) of that buffer per step (0.56 MB at N_pix=4000, N_h=16: latency-bound on
xGMI) makes sums AND counts global before the division ``sum / count`` (QFA/model.py:104), so the
result is the single-process result on the concatenated batch up to float32 re-association.
Parameters and Adam state are replicated; every rank applies the identical update.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class AccumLayout:
    npix: int
    nb: int
    nh: int

    @property
    def o_accF(self):
        return 0

    @property
    def o_sumA(self):
        return self.npix * self.nh

    @property
    def o_gPsi(self):
        return self.o_sumA + self.npix

    @property
    def o_gOmega(self):
        return self.o_gPsi + self.npix

    @property
    def o_cnt(self):
        return self.o_gOmega + self.nb

    @property
    def o_scal(self):
        return self.o_cnt + self.npix

    @property
    def size(self):
        return self.o_scal + 8


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n spectra for ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_accum(acc, group=None):
    """In-place sum of the packed buffer over the process group.  With the NCCL backend (= RCCL on
    ROCm) the device tensor goes straight to the collective; a Gloo group (CPU rigs, tests) is
    served through a host staging copy."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo" and acc.device.type != "cpu":
        host = acc.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        acc.copy_(host)
    else:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc
