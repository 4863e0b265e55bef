"""float64 oracle sums of a large batch on the host cores (test infrastructure): worker processes (spawn -- they never
touch the GPU) each run oracle.forward(return_sums=True) on a slice of arrays saved under a scratch directory."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _work(args):
    path, pfile, a, b, kw = args
    sys.path.insert(0, REPO)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    from oracle import qfa_oracle as O
    p = dict(np.load(pfile))
    arr = {k: np.load(os.path.join(path, k + ".npy"), mmap_mode="r") for k in ("delta", "error", "zabs", "mask")}
    n = b - a
    loss, g, sums, counts = O.forward(p, arr["delta"][a:b], arr["error"][a:b], arr["zabs"][a:b], arr["mask"][a:b],
                                      return_sums=True, **kw)
    return n, loss * n, sums, counts


def oracle_sums(p, batch, scratch, workers=None, chunk=64, **kw):
    """batch: dict of numpy arrays delta / error / zabs / mask.  Returns (mean loss, normalised gradients, sums, counts)."""
    import multiprocessing as mp
    os.makedirs(scratch, exist_ok=True)
    for k in ("delta", "error", "zabs", "mask"):
        np.save(os.path.join(scratch, k + ".npy"), np.ascontiguousarray(batch[k]))
    pfile = os.path.join(scratch, "params.npz")
    np.savez(pfile, **{k: np.asarray(v) for k, v in p.items()})
    B = len(batch["delta"])
    jobs = [(scratch, pfile, a, min(a + chunk, B), kw) for a in range(0, B, chunk)]
    workers = workers or max(1, min(14, (os.cpu_count() or 2) - 2, len(os.sched_getaffinity(0)) - 1))
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_work, jobs)
    tot_loss, sums, counts = 0.0, None, None
    for n, l, s, c in res:
        tot_loss += l
        if sums is None:
            sums = {k: np.array(v, dtype=np.float64) for k, v in s.items()}
            counts = {k: np.array(v, dtype=np.float64) for k, v in c.items()}
        else:
            for k in s:
                sums[k] += s[k]
                counts[k] += c[k]
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = {k: sums[k] / counts[k] for k in sums}
    return tot_loss / B, grads, sums, counts
