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


def _via_host(t, group):
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo" and t.device.type != "cpu"


def broadcast_(t, src=0, group=None):
    """In-place broadcast of one tensor from rank ``src`` (host staging under Gloo, as above)."""
    import torch.distributed as dist
    if _via_host(t, group):
        host = t.cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def all_reduce_(t, group=None):
    """In-place sum of an arbitrary tensor (the mu estimator's float64 per-pixel sums)."""
    return all_reduce_accum(t, group)


def agree_on_layout(tensors, group=None, what="replicated tensors"):
    """Raise (on every rank) unless all ranks are about to issue the SAME collectives: same number of tensors, same
    element counts.  One MAX and one MIN all-reduce of a fixed-size signature -- the list itself may differ from rank to
    rank (a rank that loaded a file carrying ``mu`` next to one that did not), and a per-tensor broadcast over lists of
    different length deadlocks instead of failing."""
    import torch
    import torch.distributed as dist
    NSLOT = 32
    if len(tensors) > NSLOT:
        raise ValueError("agree_on_layout: too many tensors")
    sig = torch.full((NSLOT + 1,), -1.0, dtype=torch.float64)
    sig[0] = len(tensors)
    for i, t in enumerate(tensors):
        sig[1 + i] = float(t.numel())
    dev = tensors[0].device if tensors else torch.device("cpu")
    if dist.get_backend(group) != "gloo" and dev.type != "cpu":
        sig = sig.to(dev)
    hi, lo = sig.clone(), sig.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    if not bool((hi == lo).all()):
        raise RuntimeError(f"data-parallel ranks disagree on the {what}: this rank has {len(tensors)} tensors of sizes "
                           f"{[int(t.numel()) for t in tensors]} (e.g. mu present on some ranks only); load the same "
                           "checkpoint on every rank before enable_data_parallel / sync_replicas")


def replicas_in_sync(tensors, group=None):
    """True when every rank holds bit-identical copies of ``tensors`` (a list).  Each rank reduces its copy to
    four float64 moments; max - min of those over the ranks must be exactly zero."""
    import torch
    import torch.distributed as dist
    sig = []
    for t in tensors:
        x = t.detach().double().reshape(-1)
        w = torch.arange(1, x.numel() + 1, dtype=torch.float64, device=x.device)
        sig += [x.sum(), (x * x).sum(), (x * w).sum(), torch.nan_to_num(x).abs().max() if x.numel() else x.sum()]
    s = torch.stack(sig) if sig else torch.zeros(1, dtype=torch.float64)
    s = torch.nan_to_num(s, nan=12345.678)
    if _via_host(s, group):
        s = s.cpu()
    hi, lo = s.clone(), s.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    return bool((hi == lo).all())


class ShardPlan(object):
    """Which rows a rank feeds into each step of an epoch under data parallelism.

    The reference's loop (QFA/model.py:204-215, QFA/dataloader.py:124-138,154-167) shuffles the whole data set
    and walks it in batches of ``batch_size``.  Here rank r owns the contiguous shard ``shard_bounds(n, r, world)``
    of the spectra (they stay where they are in that GPU's HBM); an epoch shuffles every shard with a generator
    seeded by (seed, epoch, rank) -- the same call on every rank yields the same plan, so any rank can recompute
    any other's -- and step i of the epoch takes rows ``[i*local, (i+1)*local)`` of the shuffled shard, with
    ``local = ceil(batch_size / world)``.  EVERY rank runs ``steps = ceil(max_shard / local)`` steps; a rank whose
    shard is exhausted contributes an empty batch (zeros to the all-reduce), so the collective never deadlocks on
    an uneven tail.  The union of the ranks' step-i rows is the global batch of step i (about ``batch_size`` rows).

    ``reshuffle="shard"`` (above) keeps the per-rank batch size fixed, but a global batch is always one slice from each of
    the same ``world`` sub-populations (a catalogue sorted by redshift or S/N gives every rank a different population for
    the whole run) -- a divergence from the reference's shuffle of the WHOLE set (QFA/dataloader.py:154-167).
    ``reshuffle="global"`` removes it: one permutation of all n rows per epoch, shared by the ranks; global batch k is
    ``perm[k B:(k+1) B]`` and rank r feeds the members that fall into its shard (binomially ~B / world of them; the packed
    sums and COUNTS are all-reduced, so uneven contributions need nothing special).  The batches are then exactly those of a
    single process walking the same permutation; each rank's spectra still never leave its HBM.
    """

    def __init__(self, n: int, batch_size: int, rank: int = 0, world: int = 1, seed: int = 0, shuffle: bool = True,
                 reshuffle: str = "shard"):
        if world < 1 or not (0 <= rank < world):
            raise ValueError("need 0 <= rank < world")
        if reshuffle not in ("shard", "global"):
            raise ValueError("reshuffle must be 'shard' or 'global'")
        self.n, self.batch_size, self.rank, self.world = int(n), int(batch_size), int(rank), int(world)
        self.seed, self.shuffle, self.reshuffle = int(seed), bool(shuffle), reshuffle
        self.lo, self.hi = shard_bounds(self.n, self.rank, self.world)
        self.local = -(-self.batch_size // self.world)
        max_shard = -(-self.n // self.world)
        if reshuffle == "global":
            # the reference's own walk: ceil(n / batch_size) global batches per epoch (QFA/dataloader.py:124-138)
            self.steps = -(-self.n // self.batch_size) if self.n > 0 else 0
        else:
            self.steps = -(-max_shard // self.local) if self.n > 0 else 0

    def epoch_rows(self, epoch: int, rank=None):
        """list (one entry per step, possibly empty arrays) of GLOBAL row indices for ``rank`` (default: own)"""
        import numpy as np
        r = self.rank if rank is None else int(rank)
        lo, hi = shard_bounds(self.n, r, self.world)
        if self.reshuffle == "global":
            # ONE permutation of the whole data set per epoch, the same on every rank (seed, epoch -- not the rank): global
            # batch k is perm[k B : (k + 1) B], exactly the batches a single process walking this permutation would form
            # (reference QFA/dataloader.py:154-167), and a rank contributes the members that live in its resident shard
            # -- about B / world of them, B over all ranks.  Indices only: nothing moves between GPUs.
            perm = np.arange(self.n)
            if self.shuffle:
                np.random.default_rng([self.seed, int(epoch)]).shuffle(perm)
            out = []
            for k in range(self.steps):
                rows = perm[k * self.batch_size:(k + 1) * self.batch_size]
                out.append(rows[(rows >= lo) & (rows < hi)])
            return out
        rows = np.arange(lo, hi)
        if self.shuffle:
            np.random.default_rng([self.seed, int(epoch), r]).shuffle(rows)
        return [rows[i * self.local:(i + 1) * self.local] for i in range(self.steps)]

    def global_batch(self, epoch: int, step: int):
        """rows of the whole global batch of a step (the union over the ranks), in the order a single process would walk them"""
        import numpy as np
        if self.reshuffle == "global":
            perm = np.arange(self.n)
            if self.shuffle:
                np.random.default_rng([self.seed, int(epoch)]).shuffle(perm)
            return perm[step * self.batch_size:(step + 1) * self.batch_size]
        return np.concatenate([self.epoch_rows(epoch, rank=r)[step] for r in range(self.world)])
