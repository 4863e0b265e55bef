"""N > 1 path.  CPU (gloo, world_size 2): the data-parallel protocol -- shard, accumulate raw sums
and counts per rank, ONE all-reduce of the packed buffer, then sum/count -- reproduces the
single-process result, including the NaN and red-only count semantics.  The per-rank accumulation
is played by the CPU oracle here (no GPU in this container); the packing, the collective helper and
the shard arithmetic are the product's (qfa_amd/distributed.py).
GPU (-m gpu): two ranks sharing cuda:0 over gloo run the real HIP path end to end."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")


def _case():
    from qfa_amd import synthetic
    wav, nb, nr = synthetic.wavelength_grid(220)
    p, mu = synthetic.mock_parameters(220, nb, 4, seed=8)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 7, seed=81, red_only=(5,), dead_range=(100, 104))
    return p, mu, nb, b


def _pack(lay, sums, counts, nll_sum, n):
    acc = np.zeros(lay.size, dtype=np.float64)
    F = None
    acc[lay.o_gPsi:lay.o_gPsi + lay.npix] = sums["Psi"]
    acc[lay.o_gOmega:lay.o_gOmega + lay.nb] = sums["omega"]
    acc[lay.o_cnt:lay.o_cnt + lay.npix] = counts["Psi"]
    acc[lay.o_accF:lay.o_accF + lay.npix * lay.nh] = sums["F"].ravel()   # test packs gF itself (sumA = 0)
    acc[lay.o_scal + 0] = sums["tau0"]
    acc[lay.o_scal + 1] = sums["c0"]
    acc[lay.o_scal + 2] = sums["beta"]
    acc[lay.o_scal + 3] = counts["tau0"]
    acc[lay.o_scal + 4] = nll_sum
    acc[lay.o_scal + 5] = n
    return acc


def _normalise(lay, acc):
    with np.errstate(invalid="ignore", divide="ignore"):
        cnt = acc[lay.o_cnt:lay.o_cnt + lay.npix]
        g = {"F": acc[:lay.npix * lay.nh].reshape(lay.npix, lay.nh) / cnt[:, None],
             "Psi": acc[lay.o_gPsi:lay.o_gPsi + lay.npix] / cnt,
             "omega": acc[lay.o_gOmega:lay.o_gOmega + lay.nb] / cnt[:lay.nb]}
        for i, k in enumerate(("tau0", "c0", "beta")):
            g[k] = acc[lay.o_scal + i] / acc[lay.o_scal + 3]
        return acc[lay.o_scal + 4] / acc[lay.o_scal + 5], g


def _worker_cpu(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import qfa_oracle as O
    from qfa_amd.distributed import AccumLayout, all_reduce_accum, shard_bounds
    p, mu, nb, b = _case()
    lay = AccumLayout(220, nb, 4)
    lo, hi = shard_bounds(7, rank, world)
    _, _, sums, counts = O.forward(p, b["delta"][lo:hi], b["error"][lo:hi], b["zabs"][lo:hi], b["mask"][lo:hi],
                                   return_sums=True)
    nll = sum(O.nll_and_grads_single(p, b["delta"][s], b["error"][s], b["zabs"][s], b["mask"][s])[0]
              for s in range(lo, hi))
    acc = torch.tensor(_pack(lay, sums, counts, nll, hi - lo))
    all_reduce_accum(acc)
    loss, g = _normalise(lay, acc.numpy())
    if rank == 0:
        q.put((loss, {k: np.asarray(v) for k, v in g.items()}))
    dist.destroy_process_group()


def test_dp_protocol_world2_gloo_cpu():
    from oracle import qfa_oracle as O
    from qfa_amd.distributed import AccumLayout, shard_bounds
    assert [shard_bounds(7, r, 2) for r in range(2)] == [(0, 4), (4, 7)]
    assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert AccumLayout(1913, 720, 8).size == 1913 * 8 + 3 * 1913 + 720 + 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_cpu, args=(r, 2, port, q)) for r in range(2)]
    [pr.start() for pr in procs]
    loss, g = _collect(procs, q, 120)
    [pr.join(60) for pr in procs]
    assert all(pr.exitcode == 0 for pr in procs)
    p, mu, nb, b = _case()
    oloss, og = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    assert abs(loss - oloss) < 1e-9 * abs(oloss)
    for k in KEYS:
        ref = np.asarray(og[k])
        assert np.array_equal(np.isnan(g[k]), np.isnan(ref)), k
        ok = ~np.isnan(ref)
        assert np.allclose(g[k][ok], ref[ok], rtol=1e-9, atol=1e-12), k
    assert np.isnan(og["Psi"][100:104]).all()


def _worker_gpu(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qfa_amd import QFA, Adam, step_scheduler
    from qfa_amd.distributed import shard_bounds
    dev = torch.device("cuda:0")
    p, mu, nb, b = _case()
    m = QFA(nb, 220 - nb, 4, dev, model_params=p)
    m.enable_data_parallel()
    lo, hi = shard_bounds(7, rank, world)
    T = lambda x: torch.tensor(x[lo:hi], device=dev)
    opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)
    loss, g = m.forward(T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"]))
    m.step(opt, T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"]))
    out = (loss.item(), {k: v.cpu().numpy() for k, v in g.items()}, {k: v.cpu().numpy() for k, v in m.parameters.items()})
    gathered = [None, None]
    dist.all_gather_object(gathered, out[2])
    if rank == 0:
        q.put((out, gathered))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_dp_two_ranks_hip_path_matches_single_process():
    from oracle import qfa_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_gpu, args=(r, 2, port, q)) for r in range(2)]
    [pr.start() for pr in procs]
    (loss, g, newp), gathered = _collect(procs, q, 300)
    [pr.join(60) for pr in procs]
    assert all(pr.exitcode == 0 for pr in procs)
    p, mu, nb, b = _case()
    oloss, og = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    assert abs(loss - oloss) / abs(oloss) < 1e-5
    for k in KEYS:
        ref = np.asarray(og[k])
        assert np.array_equal(np.isnan(g[k]), np.isnan(ref)), k
        ok = ~np.isnan(ref)
        assert np.linalg.norm(g[k][ok] - ref[ok]) / np.linalg.norm(ref[ok]) < 2e-4, k
    for k in KEYS:       # replicas stay identical: both ranks applied the same update
        assert np.array_equal(gathered[0][k], gathered[1][k], equal_nan=True), k


# ------------------------------------------------------------------------------------------------
# Data-parallel TRAINING: shard plan + loop (reference QFA/model.py:204-215, QFA/dataloader.py:124-138,154-167)
# ------------------------------------------------------------------------------------------------
def test_shard_plan_covers_every_row_once_and_equalises_steps():
    from qfa_amd.distributed import ShardPlan, shard_bounds
    for n, bs, world in ((23, 8, 2), (7, 4, 4), (100, 16, 8), (5, 64, 2), (16, 5, 3)):
        plans = [ShardPlan(n, bs, r, world, seed=11) for r in range(world)]
        assert len({p.steps for p in plans}) == 1                      # same number of steps on every rank
        for epoch in range(3):
            rows = [p.epoch_rows(epoch) for p in plans]
            assert all(len(r) == plans[0].steps for r in rows)
            seen = np.concatenate([np.concatenate(r) if len(r) else np.zeros(0, int) for r in rows])
            assert sorted(seen.tolist()) == list(range(n))             # a permutation of the data set
            for r, p in enumerate(plans):                              # a rank only ever touches its own shard
                lo, hi = shard_bounds(n, r, world)
                mine = np.concatenate(rows[r]) if len(rows[r]) else np.zeros(0, int)
                assert ((mine >= lo) & (mine < hi)).all()
                assert [list(x) for x in plans[0].epoch_rows(epoch, rank=r)] == [list(x) for x in rows[r]]
            for step in range(plans[0].steps):                         # global batch of a step ~ batch_size rows
                assert sum(len(rows[r][step]) for r in range(world)) <= world * plans[0].local
        e0, e1 = plans[0].epoch_rows(0), plans[0].epoch_rows(1)
        if n >= 16:
            assert [list(x) for x in e0] != [list(x) for x in e1]      # reshuffled every epoch


def test_shard_plan_global_reshuffle_forms_the_single_process_batches():
    """ShardPlan(reshuffle="global"): one permutation of the WHOLE set per epoch (reference QFA/dataloader.py:154-167); the
    union over the ranks of step k is exactly global batch k of that permutation -- the batches of a single process -- every
    rank touches its own shard only, and all ranks run ceil(n / B) steps"""
    from qfa_amd.distributed import ShardPlan, shard_bounds
    for n, bs, world in ((23, 8, 2), (7, 4, 4), (100, 16, 8), (5, 64, 2), (16, 5, 3), (1000, 100, 8)):
        plans = [ShardPlan(n, bs, r, world, seed=11, reshuffle="global") for r in range(world)]
        assert {p.steps for p in plans} == {-(-n // bs)}
        for epoch in range(3):
            rows = [p.epoch_rows(epoch) for p in plans]
            perm = np.arange(n)
            np.random.default_rng([11, epoch]).shuffle(perm)
            for k in range(plans[0].steps):
                want = perm[k * bs:(k + 1) * bs]
                got = np.concatenate([rows[r][k] for r in range(world)])
                assert sorted(got.tolist()) == sorted(want.tolist())           # global batch k, split by residence
                assert list(plans[0].global_batch(epoch, k)) == list(want)
                for r in range(world):
                    lo, hi = shard_bounds(n, r, world)
                    assert ((rows[r][k] >= lo) & (rows[r][k] < hi)).all()
                    # (the order inside a rank's part is the permutation's)
                    assert list(rows[r][k]) == [x for x in want if lo <= x < hi]
            seen = np.concatenate([np.concatenate(r) for r in rows])
            assert sorted(seen.tolist()) == list(range(n))
        if n >= 16:
            assert [list(x) for x in plans[0].epoch_rows(0)] != [list(x) for x in plans[0].epoch_rows(1)]
    with pytest.raises(ValueError):
        ShardPlan(10, 4, 0, 2, reshuffle="nope")


def _train_case():
    from qfa_amd import synthetic
    wav, nb, nr = synthetic.wavelength_grid(96)
    p, mu = synthetic.mock_parameters(96, nb, 3, seed=21)
    # no pixel masks: with 11 spectra on 96 pixels a masked run would leave pixels unobserved in a whole batch
    # (NaN gradients, quirk Q3, covered elsewhere) and Adam would carry the NaN into the parameters
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 11, seed=211, masks=False, red_only=(2,))
    return p, mu, nb, b, wav


def _collect(procs, q, timeout):
    """first item of the queue, failing fast (not after `timeout`) when a worker died without answering"""
    import queue as _q
    import time as _t
    t0 = _t.time()
    while True:
        try:
            return q.get(timeout=2)
        except _q.Empty:
            if all(not pr.is_alive() for pr in procs):
                try:
                    return q.get(timeout=1)
                except _q.Empty:
                    raise AssertionError(f"workers exited without a result: {[pr.exitcode for pr in procs]}")
            if _t.time() - t0 > timeout:
                for pr in procs:
                    if pr.is_alive():
                        pr.terminate()
                raise AssertionError("workers timed out")


def _oracle_epochs(p, b, nb, plan_rows_fn, n_epochs, steps, reduce_fn):
    """the reference's loop on the oracle: per step accumulate sums/counts of `plan_rows_fn(epoch, step)` rows,
    reduce_fn(packed) (the all-reduce, or identity), normalise, Adam + clip; i advances per epoch (quirk Q4)"""
    from oracle import qfa_oracle as O
    from qfa_amd.distributed import AccumLayout
    lay = AccumLayout(96, nb, 3)
    params = {k: np.asarray(v, dtype=np.float64) for k, v in p.items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v = {k: np.zeros_like(x) for k, x in params.items()}
    losses = []
    for epoch in range(n_epochs):
        for step in range(steps):
            rows = plan_rows_fn(epoch, step)
            if len(rows):
                _, _, sums, counts = O.forward(params, b["delta"][rows], b["error"][rows], b["zabs"][rows],
                                               b["mask"][rows], return_sums=True)
                nll = sum(O.nll_and_grads_single(params, b["delta"][s], b["error"][s], b["zabs"][s], b["mask"][s])[0]
                          for s in rows)
                acc = _pack(lay, sums, counts, nll, len(rows))
            else:
                acc = np.zeros(lay.size)                               # an exhausted rank adds zeros
            acc = reduce_fn(acc)
            loss, g = _normalise(lay, acc)
            g = {k: np.nan_to_num(np.asarray(x)) for k, x in g.items()}   # (no dead pixel in this case)
            params, m, v = O.adam_update(m, v, epoch, params, g, O.step_lr(epoch, 1e-2, 0.9, 1), weight_decay=1e-1)
            params = O.clip_params(params)
            losses.append(loss)
    return params, losses


def _worker_train_cpu(rank, world, port, q, reshuffle="shard", bs=10):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qfa_amd.distributed import ShardPlan, all_reduce_accum, replicas_in_sync
    p, mu, nb, b, wav = _train_case()
    plan = ShardPlan(11, bs, rank, world, seed=5, reshuffle=reshuffle)
    cache = {}

    def rows(epoch, step):
        if epoch not in cache:
            cache[epoch] = plan.epoch_rows(epoch)
        return cache[epoch][step]

    def reduce(acc):
        t = torch.tensor(acc)
        all_reduce_accum(t)
        return t.numpy()
    params, losses = _oracle_epochs(p, b, nb, rows, 2, plan.steps, reduce)
    same = replicas_in_sync([torch.tensor(np.asarray(params[k])) for k in KEYS])
    differ = replicas_in_sync([torch.tensor(np.asarray(params["F"]) + rank)])      # must notice a mismatch
    if rank == 0:
        q.put((params, losses, same, differ, plan.steps))
    dist.destroy_process_group()


def test_dp_training_loop_world2_gloo_cpu():
    """Two ranks over gloo run the sharded loop (uneven shards: 6 + 5 rows, local batch 5, so the last step is a
    single row on one rank and EMPTY on the other); the replicated parameters after two epochs equal a
    single-process run over the union batches."""
    from qfa_amd.distributed import ShardPlan
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_train_cpu, args=(r, 2, port, q)) for r in range(2)]
    [pr.start() for pr in procs]
    params, losses, same, differ, steps = _collect(procs, q, 300)
    [pr.join(60) for pr in procs]
    assert all(pr.exitcode == 0 for pr in procs)
    assert same and not differ
    p, mu, nb, b, wav = _train_case()
    plans = [ShardPlan(11, 10, r, 2, seed=5) for r in range(2)]
    assert steps == plans[0].steps == 2 and len(plans[1].epoch_rows(0)[1]) == 0     # the uneven tail is exercised
    union = lambda e, s: np.concatenate([pl.epoch_rows(e)[s] for pl in plans])
    ref, ref_losses = _oracle_epochs(p, b, nb, union, 2, steps, lambda a: a)
    assert np.allclose(losses, ref_losses, rtol=1e-10)
    for k in KEYS:
        assert np.allclose(params[k], ref[k], rtol=1e-9, atol=1e-12), k


def test_dp_training_loop_global_reshuffle_world2_gloo_cpu():
    """reshuffle="global": two ranks over gloo against the SINGLE-PROCESS loop of the reference over the same permutation
    (global batch k = perm[k B:(k+1) B]; batch 4 of 11 rows: a short last batch, uneven and sometimes empty rank parts)"""
    from qfa_amd.distributed import ShardPlan
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_train_cpu, args=(r, 2, port, q, "global", 4)) for r in range(2)]
    [pr.start() for pr in procs]
    params, losses, same, differ, steps = _collect(procs, q, 300)
    [pr.join(60) for pr in procs]
    assert all(pr.exitcode == 0 for pr in procs)
    assert same and not differ and steps == 3
    p, mu, nb, b, wav = _train_case()
    single = ShardPlan(11, 4, 0, 1, seed=5, reshuffle="global")          # one process, the same permutation
    parts = [ShardPlan(11, 4, r, 2, seed=5, reshuffle="global") for r in range(2)]
    sizes = [[len(x) for x in pl.epoch_rows(0)] for pl in parts]
    assert [a + c for a, c in zip(*sizes)] == [4, 4, 3] and sizes[0] != sizes[1]
    ref, ref_losses = _oracle_epochs(p, b, nb, lambda e, s: single.epoch_rows(e)[s], 2, steps, lambda a: a)
    assert np.allclose(losses, ref_losses, rtol=1e-10)
    for k in KEYS:
        assert np.allclose(params[k], ref[k], rtol=1e-9, atol=1e-12), k


def _worker_train_gpu(rank, world, port, q, backend, outdir=None, reshuffle="shard", bs=10):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from qfa_amd import QFA, Adam, step_scheduler
    from qfa_amd.dataloader import DeviceDataloader
    p, mu, nb, b, wav = _train_case()
    torch.manual_seed(100 + rank)                     # every process would draw its own F ...
    m = QFA(nb, 96 - nb, 3, dev)
    opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 1), learning_rate=1e-2, weight_decay=1e-1)
    m.enable_data_parallel(optimizer=opt)             # ... rank 0's is broadcast
    # world 1 shuffles with numpy's global generator as the reference does: unshuffled there, or the one-spectrum tail
    # batch is the red-only spectrum once in 11 epochs (no blue pixel observed: NaN gradients, quirk Q3)
    dl = DeviceDataloader(b["flux"], b["error"], b["zqso"], wav, bs, dev, rank=rank, world=world, seed=5,
                          shuffle=world > 1, reshuffle=reshuffle)
    import tempfile
    if outdir is not None:
        # ONE output directory for every rank, a checkpoint after every epoch: only the lead rank may write
        # (concurrent np.savez of one path corrupts the zip); every rank then reads the files back
        m.train(opt, dl, 2, outdir, save_interval=1, quiet=True)
        ck = os.path.join(outdir, "checkpoints")
        names = sorted(os.listdir(ck))
        assert names == ["model_parameters_epoch_01.npz", "model_parameters_epoch_02.npz"], names
        f = np.load(os.path.join(ck, names[-1]))
        assert np.array_equal(f["F"], m.F.cpu().numpy()) and np.array_equal(f["Psi"], m.Psi.cpu().numpy())
    else:
        with tempfile.TemporaryDirectory() as td:
            m.train(opt, dl, 2, td, quiet=True)
    F0 = m.F.cpu().numpy().copy()
    # a replica that drifts is caught by the check train() runs before its first step
    caught = False
    if world > 1:
        if rank == 1:
            m.F = m.F + 1.0
        try:
            m.check_replicas(opt)
        except Exception:
            caught = True
    out = ({k: v.cpu().numpy() for k, v in m.parameters.items()}, dl.mu, dl.local_size, caught)
    out[0]["F"] = F0
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


def _run_train_gpu(world, backend, outdir=None, reshuffle="shard", bs=10):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_train_gpu, args=(r, world, port, q, backend, outdir, reshuffle, bs)) for r in range(world)]
    [pr.start() for pr in procs]
    gathered = _collect(procs, q, 300)
    [pr.join(120) for pr in procs]
    assert all(pr.exitcode == 0 for pr in procs)
    return gathered


@pytest.mark.gpu
def test_dp_train_two_ranks_sharded_loader_matches_single_process():
    """QFA.train under data parallelism with DeviceDataloader(rank, world) on two processes sharing the GPU (gloo:
    one card cannot host two RCCL ranks): replicas stay bit-identical, mu is the global mean continuum, and the
    parameters after two epochs match a single-process run fed the union batches of the same shard plan."""
    from qfa_amd import QFA, Adam, step_scheduler
    from qfa_amd.dataloader import DeviceDataloader
    from qfa_amd.distributed import ShardPlan
    import tempfile
    with tempfile.TemporaryDirectory() as shared:            # both ranks train into the SAME output directory
        g = _run_train_gpu(2, "gloo", shared)
    (p0, mu0, n0, c0), (p1, mu1, n1, c1) = g
    assert (n0, n1) == (6, 5) and c0 and c1
    for k in KEYS:
        assert np.array_equal(p0[k], p1[k], equal_nan=True), k
    assert np.array_equal(mu0, mu1)
    # single process, same global batches
    dev = torch.device("cuda:0")
    p, mu, nb, b, wav = _train_case()
    torch.manual_seed(100)
    m = QFA(nb, 96 - nb, 3, dev)
    opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 1), learning_rate=1e-2, weight_decay=1e-1)
    dl = DeviceDataloader(b["flux"], b["error"], b["zqso"], wav, 10, dev, shuffle=False)
    assert np.allclose(dl.mu, mu0, rtol=1e-12)
    m.mu = torch.tensor(dl.mu, dtype=torch.float32, device=dev)
    plans = [ShardPlan(11, 10, r, 2, seed=5) for r in range(2)]
    assert len(plans[1].epoch_rows(0)[1]) == 0                   # rank 1 ran an empty step (zeros to the all-reduce)
    for epoch in range(2):
        for step in range(plans[0].steps):
            rows = np.concatenate([pl.epoch_rows(epoch)[step] for pl in plans])
            d, e, z, mk = dl._build(rows)
            m.step(opt, d, e, z, mk)
        opt.step()
    for k in KEYS:
        a, r = p0[k].astype(np.float64), m.parameters[k].cpu().numpy().astype(np.float64)
        # the two runs differ by the order of float32 sums only; Adam's first steps move every element by ~lr * sign(g),
        # so an element whose gradient sits at rounding level may land 2 lr apart: loose bound, exactness of the
        # protocol is test_dp_training_loop_world2_gloo_cpu's job
        assert np.isfinite(a).all() and np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-30) < 1e-2, k


@pytest.mark.gpu
def test_dp_train_two_ranks_global_reshuffle_matches_the_single_process_batches():
    """DeviceDataloader(reshuffle="global") under QFA.train on two processes sharing the GPU: every rank feeds the members of
    global batch k = perm[k B:(k+1) B] that live in its resident shard (uneven parts, through the indexed form); against a
    single process walking the SAME permutation -- the reference's batches (QFA/dataloader.py:154-167)"""
    from qfa_amd import QFA, Adam, step_scheduler
    from qfa_amd.dataloader import DeviceDataloader
    from qfa_amd.distributed import ShardPlan
    g = _run_train_gpu(2, "gloo", None, "global", 4)
    (p0, mu0, n0, c0), (p1, mu1, n1, c1) = g
    assert (n0, n1) == (6, 5) and c0 and c1
    for k in KEYS:
        assert np.array_equal(p0[k], p1[k], equal_nan=True), k
    dev = torch.device("cuda:0")
    p, mu, nb, b, wav = _train_case()
    torch.manual_seed(100)
    m = QFA(nb, 96 - nb, 3, dev)
    opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 1), learning_rate=1e-2, weight_decay=1e-1)
    dl = DeviceDataloader(b["flux"], b["error"], b["zqso"], wav, 4, dev, shuffle=False)
    m.mu = torch.tensor(dl.mu, dtype=torch.float32, device=dev)
    single = ShardPlan(11, 4, 0, 1, seed=5, reshuffle="global")
    assert single.steps == 3
    for epoch in range(2):
        for rows in single.epoch_rows(epoch):
            m.step(opt, *dl._build(rows))
        opt.step()
    for k in KEYS:
        a, r = p0[k].astype(np.float64), m.parameters[k].cpu().numpy().astype(np.float64)
        assert np.isfinite(a).all() and np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-30) < 1e-2, k


@pytest.mark.gpu
def test_rccl_path_world1_rehearsal():
    """The RCCL (backend "nccl") collective path on the one card this box has: world size 1, so the all-reduce and
    the broadcasts are real RCCL calls on device tensors (no host staging) even though nothing is exchanged."""
    g = _run_train_gpu(1, "nccl")
    (p0, mu0, n0, c0), = g
    assert n0 == 11 and all(np.isfinite(p0[k]).all() for k in KEYS)


def _worker_layout(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qfa_amd.distributed import agree_on_layout
    same = [torch.zeros(5), torch.zeros(0), torch.zeros(3)]
    agree_on_layout(same)                                            # identical lists: passes on every rank
    mine = same + ([torch.zeros(7)] if rank == 1 else [])           # rank 1 also holds a "mu"
    try:
        agree_on_layout(mine)
        raised = False
    except RuntimeError:
        raised = True
    sizes = [torch.zeros(4 + rank)]                                  # same count, different element counts
    try:
        agree_on_layout(sizes)
        raised2 = False
    except RuntimeError:
        raised2 = True
    dist.barrier()                                                   # nobody is stuck in a collective
    q.put((rank, raised, raised2))
    dist.destroy_process_group()


def test_replica_tensor_lists_that_differ_raise_on_every_rank_instead_of_deadlocking():
    """sync_replicas / check_replicas agree on the tensor list first: a rank that holds mu next to one that does
    not, or tensors of different sizes, raises on EVERY rank; a per-tensor broadcast would hang instead."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_layout, args=(r, 2, port, q)) for r in range(2)]
    [pr.start() for pr in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [pr.join(60) for pr in procs]
    assert all(pr.exitcode == 0 for pr in procs)
    assert res == [(0, True, True), (1, True, True)]
