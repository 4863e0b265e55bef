"""The bench's full c3 batch (100 000 x 4000, N_h = 16, masks; the seeds of bench.py) against the float64 oracle summed over the SAME
spectra on the host cores (tools/oracle_pool.py) -- VERDICT r4 weak 1(c): until round 5 the oracle met the automatic k_grads_t
dispatch at 24 613 spectra only.  One HIP launch per input form (zabs kernels, factored-z kernels): normalised gradients, loss,
per-spectrum NLL.  GPU box:  python tools/c3_100k_vs_oracle.py [B] > gpurun_out/r5_c3_100k_vs_oracle.txt"""
import os, sys, tempfile, time

if __name__ == "__main__":
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, REPO)
    import numpy as np
    import torch
    from qfa_amd import QFA, synthetic
    import qfa_amd.model as M
    from tools import oracle_pool
    from tools import parity_sections as PS
    M.AUTO_FACTOR_ZABS = False
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    SLAB0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # first 25 000-spectrum slab of the bench's batch to use
    CHUNK = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # > 0: launches of CHUNK spectra, packed buffers summed in float64
    npix, nh = 4000, 16
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    parts = [synthetic.make_batch_torch(p, mu, wav, nb, min(25000, B - s0), 20220700 + 3 + 17 * (i + SLAB0), dev, masks=True, return_zq=True)
             for i, s0 in enumerate(range(0, B, 25000))]
    batch = tuple(torch.cat([q[j] for q in parts]) for j in range(4))
    zfac = ((1.0 + torch.cat([q[4] for q in parts])).contiguous(), torch.tensor((wav[:nb] / synthetic.LYA).astype(np.float32), device=dev))
    del parts
    host = {k: x.cpu().numpy() for k, x in zip(("delta", "error", "zabs", "mask"), batch)}
    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        ol, og, sums, counts = oracle_pool.oracle_sums(p, host, td)
    print(f"float64 oracle over {B} spectra on the host cores: {time.time() - t0:.0f} s; loss {ol:.9f}")
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64)[ok] - np.asarray(b, np.float64)[ok]) / np.linalg.norm(np.asarray(b, np.float64)[ok]))
    for name, zf in (("zabs kernels", None), ("factored-z kernels", zfac)):
        m = QFA(nb, nr, nh, dev, model_params=p)
        m.mu = torch.tensor(mu, device=dev)
        nll = torch.empty(B, device=dev)
        if CHUNK <= 0:
            acc = m.accumulate(batch[0], batch[1], batch[2] if zf is None else None, batch[3], nll=nll, zfac=zf).clone()
        else:
            tot = None
            for a in range(0, B, CHUNK):
                b_ = min(a + CHUNK, B)
                part = m.accumulate(batch[0][a:b_], batch[1][a:b_], batch[2][a:b_] if zf is None else None, batch[3][a:b_], nll=nll[a:b_],
                                    zfac=None if zf is None else (zf[0][a:b_], zf[1]))
                tot = part.double() if tot is None else tot + part.double()
            acc = tot.float()
            name = f"{name} in launches of {CHUNK}"
        loss, g = m._finalize(acc, True)
        out = {}
        for k in ("F", "Psi", "omega"):
            ref = np.asarray(og[k]); ok = ~np.isnan(ref)
            out[k] = rel(g[k].cpu().numpy(), ref)
        for k in ("tau0", "c0", "beta"):
            out[k] = abs(g[k].item() - float(og[k])) / abs(float(og[k]))
        out["loss"] = abs(loss.item() - ol) / abs(ol)
        # the yardstick of the cancellation regime (tests/test_stage3_precision.py): gF = F sumA - accF, both sums ~ `terms`
        n = npix * nh
        accF = acc[:n].double().cpu().numpy().reshape(npix, nh)
        cnt = acc[n + 2 * npix + nb: n + 3 * npix + nb].double().cpu().numpy()
        terms = np.linalg.norm(accF / np.maximum(cnt, 1.0)[:, None])
        dF = g["F"].cpu().numpy().astype(np.float64) - np.asarray(og["F"])
        out["F_over_terms"] = float(np.linalg.norm(dF) / terms)
        out["cancellation"] = float(terms / np.linalg.norm(np.asarray(og["F"])))
        print(f"{name:20s} one launch of {B}: " + "  ".join(f"{k} {v:.3e}" for k, v in out.items()))
