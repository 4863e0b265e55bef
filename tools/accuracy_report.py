"""Print relative errors of the HIP path against the float64 oracle (GPU box only)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qfa_oracle as O
from qfa_amd import QFA, synthetic

dev = torch.device("cuda:0")
T = lambda x: torch.tensor(x, device=dev)
KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    ok = ~np.isnan(b)
    return np.linalg.norm(a[ok] - b[ok]) / np.linalg.norm(b[ok])

def case(name, p, mu, wav, nb, B, seed, **kw):
    b = synthetic.make_batch_numpy(p, mu, wav, nb, B, seed=seed, **kw)
    m = QFA(nb, len(wav) - nb, p["F"].shape[1], dev, model_params=p); m.mu = T(mu)
    nll = torch.empty(B, device=dev)
    acc = m.accumulate(T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"]), nll=nll)
    loss, g = m._finalize(acc, True)
    ol, og = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    l32, g32 = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"], dtype=np.float32)
    per = np.array([O.nll_and_grads_single(p, b["delta"][s], b["error"][s], b["zabs"][s], b["mask"][s])[0] for s in range(B)])
    print(f"{name}: loss rel {abs(loss.item()-ol)/abs(ol):.2e} (np32 {abs(l32-ol)/abs(ol):.2e})  per-spectrum nll max rel {np.max(np.abs(nll.cpu().numpy()-per)/np.abs(per)):.2e}")
    print("   " + "  ".join(f"{k} {rel(g[k].cpu().numpy(), og[k]):.1e}({rel(g32[k], og[k]):.1e})" for k in KEYS))
    ll, hm, hc, cont, unc = [x.cpu().numpy() for x in m.predict(T(b["flux"]), T(b["error"]), T(b["zabs"]), T(b["mask"]))]
    e = np.zeros(5)
    for s in range(min(B, 8)):
        o = O.predict_single(p, mu, b["flux"][s], b["error"][s], b["zabs"][s], b["mask"][s])
        e = np.maximum(e, [abs(ll[s]-o[0])/abs(o[0]), rel(hm[s], o[1]), rel(hc[s], o[2]), np.max(np.abs(cont[s]-o[3]))/np.max(np.abs(o[3])), rel(unc[s], o[4])])
    print("   predict: ll %.1e hmean %.1e hcov %.1e cont(maxabs/max) %.1e unc %.1e" % tuple(e))

from tests.conftest import GOLDEN
p, mu = O.load_params_npz(os.path.join(GOLDEN, "model_parameters.npz"))
wav, nb, nr = synthetic.wavelength_grid()
print("lib:", os.environ.get("QFA_HIP_LIB", "default"))
case("sdss k8 B=8 (G4)", p, mu, wav, nb, 8, 20220704, red_only=(3,), dead_range=(900, 910))
case("sdss k8 B=64", p, mu, wav, nb, 64, 1)
for npix, nh, seed in ((2000, 8, 2), (4000, 16, 3), (640, 16, 13)):
    w, b_, _ = synthetic.wavelength_grid(npix)
    pp, mm = synthetic.mock_parameters(npix, b_, nh, seed=seed)
    case(f"mock npix={npix} k={nh} B=24", pp, mm, w, b_, 24, 200 + seed)
