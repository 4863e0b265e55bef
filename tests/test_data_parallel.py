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
    loss, g = q.get(timeout=120)
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
    (loss, g, newp), gathered = q.get(timeout=300)
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
