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

The resident, indexed input form (ABI v3, include/qfa_hip.h ``qfa_batch_t.rows`` / ``row_stride``).  The reference
recomputes ``delta`` and re-uploads four arrays for EVERY batch (QFA/dataloader.py:124-138) although delta depends on
``mu`` and ``tau`` only, and shuffles an epoch by permuting the whole data set on the host (:154-167).  Here ``delta`` and the
mask are built ONCE (``qfa_build_resident_f32``; again in ``set_tau``), every row padded to a multiple of 32 pixels (128-byte
row starts), and an epoch is one permutation uploaded by ``rewind()``: ``next_batch_rows()`` returns a ``ResidentBatch`` --
the resident arrays plus a slice of that permutation -- which ``QFA.step(batch=...)`` hands to the kernels as it is.  No
per-step kernel, copy or host transfer; ``QFA.train`` takes this path by itself.  ``next_batch()`` still materialises the
reference's 4-tuple for callers that want tensors.

Data parallelism (no counterpart in the reference): ``DeviceDataloader(..., rank=r, world=w,
seed=s)`` keeps only rank r's contiguous shard of the spectra in HBM, draws the epoch order from
``qfa_amd.distributed.ShardPlan`` (shared seed, disjoint shards, the SAME number of steps on every rank -- an
exhausted rank returns empty batches, which ``QFA.forward`` turns into zeros for the all-reduce) and estimates
``mu`` from the all-reduced per-pixel sums, so every rank trains against the same mean continuum.
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
                 shuffle=True, mode="train", paths=None, rank=0, world=1, seed=0, group=None, sharded_input=False,
                 global_size=None, reshuffle="shard"):
        """``rank`` / ``world`` / ``seed`` / ``group``: data parallelism (see the module docstring).  By default the
        arrays hold ALL spectra and the loader keeps rows ``shard_bounds(N, rank, world)``; with
        ``sharded_input=True`` they already are this rank's shard of ``global_size`` spectra.  ``reshuffle``: "shard" (every
        epoch shuffles inside the rank's shard) or "global" (one permutation of the whole set per epoch, every rank feeds the
        members of a global batch that live in its shard: the reference's batches; qfa_amd.distributed.ShardPlan)."""
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
        from .distributed import ShardPlan, shard_bounds
        self.rank, self.world, self.group = int(rank), int(world), group
        # flux / error: host arrays (the reference's case), or float32 tensors that already live on the device (no host copy)
        on_dev = isinstance(flux, torch.Tensor)
        if on_dev:
            if not isinstance(error, torch.Tensor) or flux.dtype != f32 or error.dtype != f32 or flux.device != self.device \
                    or error.device != self.device:
                raise _lib.QFAHipError("flux / error tensors must both be float32 on the loader's device")
            zqso = zqso.detach().cpu().numpy() if isinstance(zqso, torch.Tensor) else zqso
        else:
            flux, error = np.asarray(flux, dtype=np.float32), np.asarray(error, dtype=np.float32)
        zqso = np.asarray(zqso, dtype=np.float64).reshape(-1)
        if flux.shape != error.shape or flux.ndim != 2 or flux.shape[1] != self.Npix or len(zqso) != flux.shape[0]:
            raise _lib.QFAHipError("flux / error must be (N, len(wav_grid)) and zqso (N,)")
        paths = np.asarray(paths) if paths is not None else None
        if self.world > 1 and sharded_input:
            if global_size is None:
                raise ValueError("sharded_input=True needs global_size (the number of spectra over all ranks)")
            n_global = int(global_size)
            lo, hi = shard_bounds(n_global, self.rank, self.world)
            if hi - lo != flux.shape[0]:
                raise ValueError(f"rank {rank} must hold rows [{lo}, {hi}) of the data set, got {flux.shape[0]} rows")
        else:
            n_global = int(flux.shape[0])
            lo, hi = shard_bounds(n_global, self.rank, self.world)
            flux, error, zqso = flux[lo:hi], error[lo:hi], zqso[lo:hi]
            paths = paths[lo:hi] if paths is not None else None
        self._row0 = lo                                         # global index of the first resident row
        # resident storage: rows padded to a multiple of 32 pixels, so that every row starts on a 128-byte line (the kernels
        # move 128-byte row segments; N_pix = 1913 / 9243 -- the reference's two models -- are no multiple of 32)
        self._stride = (self.Npix + 31) // 32 * 32
        def padded(a):
            n = a.shape[0]
            a = a if on_dev else torch.as_tensor(np.ascontiguousarray(a), device=self.device)
            if self._stride == self.Npix:
                return a.contiguous()
            t = torch.zeros((n, self._stride), dtype=f32, device=self.device)
            t[:, :self.Npix] = a
            return t
        self._flux_pad, self._error_pad = padded(flux), padded(error)
        self.flux, self.error = self._flux_pad[:, :self.Npix], self._error_pad[:, :self.Npix]      # (views)
        self.zqso = zqso
        self.data_size = n_global                               # what QFA.train divides by (model.py:205)
        self.local_size = int(self.flux.shape[0])
        self.pathlist = paths if paths is not None else np.arange(lo, hi)
        self._zq_dev = torch.as_tensor(self.zqso, device=self.device)
        self._wav_dev = torch.as_tensor(self.wav_grid, device=self.device)
        # factored-z input form (include/qfa_hip.h): 1 + zabs[s][i] = (1 + z_qso[s]) wav_i / 1215.67 (reference
        # QFA/dataloader.py:102), so the kernels can take the two factors instead of reading the (B, Nb) array
        self._zq1_dev = (1.0 + self._zq_dev).to(f32)
        self._pix_ratio = torch.as_tensor((self.wav_grid[:self.Nb] / LYA).astype(np.float32), device=self.device)
        self.factored_z = True                                  # attach the factors to the zabs tensor of every batch
        self._plan = ShardPlan(n_global, self.batch_size, self.rank, self.world, seed, shuffle, reshuffle) if self.world > 1 else None
        self._epoch = -1
        self._steps = None                                      # DP: list of local row arrays of the current epoch
        self._order = np.arange(self.local_size)
        self._order_dev = None                                  # the epoch's row order on the device (int32), uploaded by rewind()
        self._step_off = None                                   # DP: offsets of the steps inside _order_dev
        self._prefetched = False                                # the next epoch's order is already drawn (prefetch_epoch)
        self._sort_batches = True                               # rows of a batch in storage order (see _upload_order; property below)
        self.cur = 0
        self._window = int(window_length_for_mu)
        self._mu_raw, self._mu = self._estimate_mu(self._window)
        self._mu_dev = torch.as_tensor(self._mu, device=self.device)
        self._delta_pad = self._mask_pad = None
        self._build_resident()
        self._upload_order()

    # ------------------------------------------------------------------ from disk (row N3)
    @classmethod
    def from_files(cls, paths, wav_grid, batch_size, device, nprocs=1, **kw):
        """spectra read from per-spectrum .npz files (reference QFA/dataloader.py:18-45,84-88)"""
        from . import io
        world, rank = int(kw.get("world", 1)), int(kw.get("rank", 0))
        if world > 1:                                          # every rank reads its own shard of the files only
            from .distributed import shard_bounds
            paths = list(paths)
            lo, hi = shard_bounds(len(paths), rank, world)
            kw.update(sharded_input=True, global_size=len(paths))
            paths = paths[lo:hi]
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
        """reference QFA/dataloader.py:109-111; under data parallelism the per-pixel sums of the shards are
        all-reduced between the two halves (qfa_mu_sums_f64 / qfa_mu_finish_f64)."""
        h, st = _lib.lib(), _lib.current_stream(self.device)
        scratch = torch.zeros(2 * self.Npix, dtype=torch.float64, device=self.device)
        raw = torch.empty(self.Npix, dtype=torch.float64, device=self.device)
        sm = torch.empty(self.Npix, dtype=torch.float64, device=self.device)
        if self.local_size > 0:
            _lib.check(h.qfa_mu_sums_f64(
                C.c_void_p(self._flux_pad.data_ptr()), C.c_void_p(self._error_pad.data_ptr()),
                C.c_void_p(self._zq_dev.data_ptr()), C.c_void_p(self._wav_dev.data_ptr()), float(self.wav_grid[0]), self._which,
                self.local_size, self.Npix, self.Nb, self._stride, C.c_void_p(scratch.data_ptr()), st), "qfa_mu_sums_f64")
        if self.world > 1:
            from .distributed import all_reduce_
            all_reduce_(scratch, self.group)
        _lib.check(h.qfa_mu_finish_f64(C.c_void_p(scratch.data_ptr()), self.Npix, window, C.c_void_p(raw.data_ptr()),
                                       C.c_void_p(sm.data_ptr()), st), "qfa_mu_finish_f64")
        return raw.cpu().numpy(), sm.cpu().numpy()

    @property
    def mu(self):
        return self._mu

    # ------------------------------------------------------------------ resident form
    def _build_resident(self):
        """delta = flux - mu exp(-tau_total) and the mask of EVERY resident row, once (reference QFA/dataloader.py:29,135-136
        per batch); called again when mu / tau change (set_tau)."""
        n = self.local_size
        self._delta_pad = torch.empty((n, self._stride), dtype=f32, device=self.device)
        self._mask_pad = torch.empty((n, self._stride), dtype=torch.bool, device=self.device)
        self._zq1_res = torch.empty((n,), dtype=f32, device=self.device)
        if n == 0:
            return
        _lib.check(_lib.lib().qfa_build_resident_f32(
            C.c_void_p(self._flux_pad.data_ptr()), C.c_void_p(self._error_pad.data_ptr()), C.c_void_p(self._zq_dev.data_ptr()),
            C.c_void_p(self._wav_dev.data_ptr()), float(self.wav_grid[0]), C.c_void_p(self._mu_dev.data_ptr()), self._which,
            n, self.Npix, self.Nb, self._stride, C.c_void_p(self._delta_pad.data_ptr()), C.c_void_p(self._mask_pad.data_ptr()),
            C.c_void_p(self._zq1_res.data_ptr()), _lib.current_stream(self.device)), "qfa_build_resident_f32")

    @property
    def sort_batches(self):
        return self._sort_batches

    @sort_batches.setter
    def sort_batches(self, on):
        """(the device copy of the epoch's order is what carries the sorting: a change re-uploads it, so that next_batch() and
        next_batch_rows() keep describing the same rows)"""
        on = bool(on)
        if on != self._sort_batches:
            self._sort_batches = on
            if getattr(self, "_order_dev", None) is not None:
                self._upload_order()

    def _upload_order(self):
        """The epoch's row order as ONE int32 device array (batches are contiguous slices of it).  ``sort_batches`` (default):
        the rows of every batch ascending.  A batch is a SET -- loss and gradients are sums over it (reference
        QFA/model.py:98-104), which spectra share a batch is what the shuffle decides -- and walking its rows in storage
        order keeps the 16 rows a wave streams within a few address-translation pages instead of 16 random ones per array:
        at c3 (4 x 10^5 resident rows of 16 KB) a step on rows in shuffled order runs 6 % slower than on the same rows
        sorted (pass 1 + 9 %, pass 2 + 5 %; bench.py `epoch.indexed_step`).  ``next_batch()`` keeps the reference's order."""
        if self._plan is not None and self._steps is not None:
            lens = [len(r) for r in self._steps]
            self._step_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            steps = [np.sort(r) for r in self._steps] if self.sort_batches else self._steps
            flat = np.concatenate(steps).astype(np.int32) if sum(lens) else np.zeros((0,), dtype=np.int32)
        else:
            self._step_off = None
            flat = np.array(self._order, dtype=np.int32)
            if self.sort_batches and self.shuffle:
                nfull = len(flat) // self.batch_size
                if nfull:
                    flat[:nfull * self.batch_size].reshape(nfull, self.batch_size).sort(axis=1)
                flat[nfull * self.batch_size:].sort()
        self._order_dev = torch.as_tensor(flat, device=self.device)

    def rows_view(self, rows):
        """``ResidentBatch`` of the given int32 device tensor of LOCAL row numbers (no copy)"""
        from .resident import ResidentBatch
        return ResidentBatch(self._flux_pad, self._delta_pad, self._error_pad, self._mask_pad, self._zq1_res, self._pix_ratio,
                             rows, self.Npix, self.Nb)

    def _next_slice(self):
        """[start, end) of the next batch inside _order_dev; advances the cursor"""
        if self._plan is not None:
            if self._steps is None:
                self.rewind()
            a, b = int(self._step_off[self.cur]), int(self._step_off[self.cur + 1])
            self.cur += 1
            return a, b
        start = self.cur
        end = min(self.cur + self.batch_size, self.local_size)
        self.cur = end
        return start, end

    def next_batch_rows(self):
        """The next batch in the resident, indexed form: a ``ResidentBatch`` whose rows are a slice of the epoch's
        permutation on the device.  The same spectra ``next_batch()`` would return (reference QFA/dataloader.py:124-138), in
        storage order unless ``sort_batches`` is off; nothing is launched or copied.  Under data parallelism: this rank's
        part (possibly empty)."""
        a, b = self._next_slice()
        return self.rows_view(self._order_dev[a:b])

    def full_batches_left(self):
        """whole batches of ``batch_size`` rows left in the epoch (not under data parallelism: the parts of a rank vary)"""
        if self._plan is not None:
            return 0
        return (self.local_size - self.cur) // self.batch_size

    def next_rows_into(self, buf):
        """copy the row numbers of the next batch -- or of the next k whole batches, for a buffer of k batch_size rows -- into
        the caller's fixed int32 device buffer (a captured step graph reads it)"""
        n = int(buf.shape[0])
        if self._plan is None and n > self.batch_size:
            if n % self.batch_size or self.cur + n > self.local_size:
                raise _lib.QFAHipError(f"{n} rows asked for: not a number of whole batches left in the epoch")
            a, b = self.cur, self.cur + n
            self.cur = b
        else:
            a, b = self._next_slice()
        if b - a != n:
            raise _lib.QFAHipError(f"next batch has {b - a} rows, the buffer {n}")
        buf.copy_(self._order_dev[a:b])

    def rows_batch(self, lo, hi):
        """``ResidentBatch`` of the resident rows [lo, hi) in storage order, and their paths (the batched predict writer)"""
        lo, hi = max(int(lo), 0), min(int(hi), self.local_size)
        rows = torch.arange(lo, max(hi, lo), dtype=torch.int32, device=self.device)
        return self.rows_view(rows), self.pathlist[lo:max(hi, lo)]

    # ------------------------------------------------------------------ batches
    def _build(self, rows, out=None, idx=None):
        """rows: LOCAL row indices (into this rank's resident spectra)"""
        n = len(rows)
        if n == 0:                                            # an exhausted rank's step under data parallelism
            return (torch.empty((0, self.Npix), dtype=f32, device=self.device),
                    torch.empty((0, self.Npix), dtype=f32, device=self.device),
                    torch.empty((0, self.Nb), dtype=f32, device=self.device),
                    torch.empty((0, self.Npix), dtype=torch.bool, device=self.device))
        if idx is None:                                       # (next_batch passes a slice of the epoch's order on the device)
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
            C.c_void_p(self._flux_pad.data_ptr()), C.c_void_p(self._error_pad.data_ptr()), C.c_void_p(self._zq_dev.data_ptr()),
            C.c_void_p(idx.data_ptr()), C.c_void_p(self._wav_dev.data_ptr()), float(self.wav_grid[0]),
            C.c_void_p(self._mu_dev.data_ptr()), self._which, n, self.Npix, self.Nb, self._stride,
            C.c_void_p(delta.data_ptr()), C.c_void_p(err.data_ptr()), C.c_void_p(zabs.data_ptr()) if self.Nb > 0 else None,
            C.c_void_p(mask.data_ptr()), _lib.current_stream(self.device)), "qfa_build_batch_f32")
        if out is not None:
            # the kernel wrote the caller's tensors through raw pointers: tell torch (QFA.auto_factor_zabs keys on the version counter)
            for t in out:
                torch.autograd.graph.increment_version(t)
        if self.factored_z and out is None and self.Nb > 0:
            # QFA.forward / step / predict look for this attribute on the zabs tensor they are handed (the 4-tuple of the
            # reference's next_batch contract stays what it is; a sliced or copied zabs simply loses the attribute)
            zabs.zfac = (self._zq1_dev[idx.long()], self._pix_ratio)
        return delta, err, zabs, mask

    def have_next_batch(self):
        if self._plan is not None:
            if self._steps is None:
                self.rewind()
            return self.cur < len(self._steps)
        return self.cur < self.local_size

    def next_batch(self, out=None):
        """delta, error, zabs, mask of the next batch (reference QFA/dataloader.py:124-138); ``out`` = four
        caller-owned tensors of the batch's shape to build into.  Under data parallelism: this rank's part of the
        global batch (possibly empty)."""
        a, b = self._next_slice()
        rows = self._steps[self.cur - 1] if self._plan is not None else self._order[a:b]
        return self._build(rows, out, idx=self._order_dev[a:b] if (b > a and not self.sort_batches) else None)

    def next_batch_size(self):
        if self._plan is not None:
            return len(self._steps[self.cur]) if self._steps is not None and self.cur < len(self._steps) else 0
        return min(self.cur + self.batch_size, self.local_size) - self.cur

    def _draw_epoch(self):
        """the next epoch's row order: host shuffle (the reference's np.random.shuffle of the whole set, QFA/dataloader.py:
        154-167 -- the SAME stream of draws: np.random.seed(s) here and there give the same batches) + one upload"""
        if self._plan is not None:
            self._epoch += 1
            self._steps = [r - self._row0 for r in self._plan.epoch_rows(self._epoch)]
        elif self.shuffle:
            np.random.shuffle(self._order)
        self._upload_order()

    def prefetch_epoch(self):
        """Draw and upload the NEXT epoch's order now (``QFA.train`` calls this behind the last step of an epoch, before it
        waits for the epoch's loss): the host shuffle -- milliseconds for 10^5..10^6 rows -- then runs while the GPU still
        works through the queued steps instead of between two epochs with the GPU idle.  The following ``rewind()`` takes
        the prefetched order.  Nothing else may consume the current epoch's batches afterwards."""
        if self._prefetched:
            return
        self._draw_epoch()
        self._prefetched = True
        # the drawn order belongs to the NEXT epoch: the current one is over for next_batch() / have_next_batch() until rewind()
        self.cur = len(self._steps) if (self._plan is not None and self._steps is not None) else self.local_size

    def rewind(self):
        """shuffle and reset (reference QFA/dataloader.py:154-167); only the row order is permuted,
        the spectra stay where they are in HBM.  Data parallel: the next epoch of the shard plan."""
        if self._prefetched:
            self._prefetched = False
        else:
            self._draw_epoch()
        self.cur = 0

    def sample(self):
        """a random batch with replacement (reference QFA/dataloader.py:140-152; the reference's version cannot run:
        it asks for ``torch.tensor32``, quirk Q9)"""
        if self.local_size == 0:                            # an empty data-parallel shard
            return self._build(np.zeros((0,), dtype=np.int64))
        return self._build(np.random.randint(0, self.local_size, size=(self.batch_size,)))

    def set_device(self, device):
        """reference QFA/dataloader.py:175-179.  The spectra already live on the device given to the constructor;
        any other device is refused (there is no host path)."""
        d = torch.device(device)
        if d.type != self.device.type or (d.index is not None and d.index != self.device.index):
            raise _lib.QFAHipError(f"DeviceDataloader is resident on {self.device}; cannot serve {device}")

    def set_tau(self, tau):
        """reference QFA/dataloader.py:169-173: choose the mean optical depth (a built-in name, or
        ``functools.partial(qfa_amd.utils.tau, which=...)``); mu is re-estimated with it."""
        which = tau if isinstance(tau, str) else dict(getattr(tau, "keywords", None) or {}).get("which")
        if which not in _lib.TAU_IDS:
            raise NotImplementedError("currently available mean optical depth function: ['becker', 'fg', 'kamble']")
        self._which = _lib.TAU_IDS[which]
        self._mu_raw, self._mu = self._estimate_mu(self._window)
        self._mu_dev = torch.as_tensor(self._mu, device=self.device)
        self._build_resident()                                  # delta of every resident row follows mu and tau

    def __len__(self):
        return self.local_size

    def get_rows(self, lo, hi):
        """raw flux, error, zabs, mask, paths of the resident rows [lo, hi) in ONE launch -- what ``__getitem__``
        returns per spectrum, for a whole slice (the batched predict writer)."""
        rows = np.arange(max(int(lo), 0), min(int(hi), self.local_size))
        if len(rows) == 0:                                  # past the end, or an empty data-parallel shard
            e = lambda n, dt=f32: torch.empty((0, n), dtype=dt, device=self.device)
            return e(self.Npix), e(self.Npix), e(self.Nb), e(self.Npix, torch.bool), self.pathlist[:0]
        _, err, zabs, mask = self._build(rows)
        return self.flux[rows[0]:rows[-1] + 1], err, zabs, mask, self.pathlist[rows]

    def __getitem__(self, i):
        """raw flux (not delta), error, zabs, mask, path of spectrum i (reference dataloader.py:184-187)."""
        _, err, zabs, mask = self._build([int(i)])
        return self.flux[int(i)], err[0], zabs[0], mask[0], self.pathlist[int(i)]
