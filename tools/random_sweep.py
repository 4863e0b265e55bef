"""GPU box: randomised shapes through every input form / kernel form against the float64 oracle (a robustness sweep run by
hand; the fixed cases live in tests/).  Prints one line per case and a summary; exit code 1 on any failure."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import qfa_oracle as O
from qfa_amd import QFA, _lib, synthetic
dev = torch.device("cuda:0")
T = lambda x: torch.tensor(x, device=dev)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1234)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); ok = ~np.isnan(b)
    if not np.array_equal(np.isnan(a), np.isnan(b)): return np.inf
    n = np.linalg.norm(b[ok]); return np.linalg.norm(a[ok] - b[ok]) / n if n > 0 else np.linalg.norm(a[ok])
bad = 0
for case in range(ncase):
    nh = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 11, 13, 16, 17, 20, 24, 31, 32]))
    npix = int(rng.integers(40, 1300)) if nh <= 16 else int(rng.integers(40, 700))
    B = int(rng.choice([1, 2, 5, 15, 16, 17, 33, 63, 64, 65, 100, 130, 200] + ([300, 513, 1000] if os.environ.get("SWEEP_BIG") else [])))
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=case)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, B, seed=1000 + case, masks=bool(rng.integers(0, 2)))
    ol, og = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    zfac = (T(1.0 + b["zqso"].astype(np.float64)).float(), T((wav[:nb] / synthetic.LYA).astype(np.float32)))
    forms = [("default", 0, None), ("zfac", 0, zfac), ("det", 0, None)]
    if nh <= 16: forms += [("det+pixres", _lib.F_PASS2_PIXRES, None)]
    if nh <= 16:
        forms += [("xdl", _lib.F_PASS2_XDL, None), ("xdl+zfac", _lib.F_PASS2_XDL, zfac), ("f32", _lib.F_PASS2_F32, zfac),
                  ]
        if True:
            forms += [("pixres", _lib.F_PASS2_PIXRES, None), ("pixres+zfac", _lib.F_PASS2_PIXRES, zfac)]
    msgs = []
    for name, fl, zf in forms:
        m = QFA(nb, nr, nh, dev, model_params=p); m.mu = T(mu); m.flags = fl; m.deterministic = name.startswith("det")
        try:
            loss, g = m.forward(T(b["delta"]), T(b["error"]), T(b["zabs"]) if zf is None else None, T(b["mask"]), zfac=zf)
            torch.cuda.synchronize()
        except Exception as e:
            msgs.append(f"{name}: EXC {e}"); continue
        errs = {k: rel(g[k].cpu().numpy(), og[k]) for k in KEYS}
        le = abs(loss.item() - ol) / abs(ol)
        tolF = 3e-4 if nh > 16 else 2e-4
        ok = le < 1e-5 and errs["F"] < tolF and errs["Psi"] < 5e-5 and errs["omega"] < 5e-5 and all(errs[k] < 2e-3 for k in ("tau0", "c0", "beta"))
        if not ok: msgs.append(f"{name}: loss {le:.1e} " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    # prediction, both forms
    m = QFA(nb, nr, nh, dev, model_params=p); m.mu = T(mu)
    for name, zf in (("pred", None), ("pred+zfac", zfac)):
        ll, hm, hc, cont, unc = [x.cpu().numpy() for x in m.predict(T(b["flux"]), T(b["error"]), T(b["zabs"]) if zf is None else None, T(b["mask"]), zfac=zf)]
        s = int(rng.integers(0, B))
        o = O.predict_single(p, mu, b["flux"][s], b["error"][s], b["zabs"][s], b["mask"][s])
        e = (abs(ll[s] - o[0]) / abs(o[0]), np.max(np.abs(cont[s] - o[3])) / np.max(np.abs(o[3])), rel(unc[s], o[4]))
        if not (e[0] < 1e-5 and e[1] < 1e-4 and e[2] < 1e-4): msgs.append(f"{name}: ll {e[0]:.1e} cont {e[1]:.1e} unc {e[2]:.1e}")
    print(f"case {case:3d} npix {npix:5d} nb {nb:4d} nh {nh:2d} B {B:3d}: " + ("ok" if not msgs else " | ".join(msgs)), flush=True)
    bad += bool(msgs)
print("cases with a failure:", bad)
sys.exit(1 if bad else 0)
