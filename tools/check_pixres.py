"""GPU box: the pixel-resident form of pass 2 (k_grads_t, QFA_F_PASS2_PIXRES) against the default form and the float64 oracle
on ragged shapes, every input form, both accumulation modes.  One line per case; exit code 1 on any failure."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import qfa_oracle as O
from qfa_amd import QFA, _lib, synthetic
dev = torch.device("cuda:0")
T = lambda x: torch.tensor(x, device=dev)
KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); ok = ~np.isnan(b)
    if not np.array_equal(np.isnan(a), np.isnan(b)): return np.inf
    n = np.linalg.norm(b[ok]); return np.linalg.norm(a[ok] - b[ok]) / n if n > 0 else np.linalg.norm(a[ok])
cases = [(640, 16, 48), (640, 16, 500), (1000, 12, 700), (97, 9, 33), (2000, 16, 1500), (1913, 13, 130), (333, 16, 17),
         (4000, 16, 3000), (64, 10, 1), (1100, 11, 257), (1913, 8, 700), (97, 5, 33), (450, 1, 65), (2000, 8, 3000), (333, 7, 17)]
if len(sys.argv) > 1: cases = cases[:int(sys.argv[1])]
bad = 0
for npix, nh, B in cases:
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=npix + nh)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, B, seed=3 * npix + nh)
    ol, og = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    zfac = (T(1.0 + b["zqso"].astype(np.float64)).float(), T((wav[:nb] / synthetic.LYA).astype(np.float32)))
    msgs = []
    for name, fl, zf, det in (("x", _lib.F_PASS2_XDL, None, False), ("t", _lib.F_PASS2_XDL | _lib.F_PASS2_PIXRES, None, False),
                              ("t+zfac", _lib.F_PASS2_XDL | _lib.F_PASS2_PIXRES, zfac, False),
                              ("t+det", _lib.F_PASS2_XDL | _lib.F_PASS2_PIXRES, None, True)):
        m = QFA(nb, nr, nh, dev, model_params=p); m.mu = T(mu); m.flags = fl | _lib.F_SYNC; m.deterministic = det
        try:
            loss, g = m.forward(T(b["delta"]), T(b["error"]), T(b["zabs"]) if zf is None else None, T(b["mask"]), zfac=zf)
            torch.cuda.synchronize()
        except Exception as e:
            msgs.append(f"{name}: EXC {e}"); continue
        errs = {k: rel(g[k].cpu().numpy(), og[k]) for k in KEYS}
        le = abs(loss.item() - ol) / abs(ol)
        ok = le < 1e-5 and errs["F"] < 2e-4 and errs["Psi"] < 5e-5 and errs["omega"] < 5e-5 and all(errs[k] < 2e-3 for k in ("tau0", "c0", "beta"))
        msgs.append(("" if ok else "FAIL ") + f"{name}: " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
        bad += not ok
    print(f"npix {npix:5d} nb {nb:4d} nh {nh:2d} B {B:4d}:\n   " + "\n   ".join(msgs), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
