"""Achieved errors of the shipped HIP build against the float64 oracle (GPU box only) -> profiles/r<N>_accuracy.txt.

Small batches (B = 24) at the four shapes (1913, 8), (2000, 8), (4000, 16), (8000, 32): loss, per-spectrum NLL, the six
normalised gradients (in brackets: the float32 numpy oracle against the float64 one, i.e. what float32 arithmetic in a
different summation order costs), prediction outputs.  With --full: the section-by-section errors of one launch at
BASELINE's full sizes against chunked float64 sums and an oracle sub-batch (tools/parity_sections.py), the numbers
tests/test_full_size_parity.py asserts on.
"""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import qfa_oracle as O
from qfa_amd import QFA, synthetic
from tools import parity_sections as PS

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
    print("   predict: ll %.1e hmean %.1e hcov %.1e cont(maxabs/max) %.1e unc %.1e" % tuple(e), flush=True)

def full(npix, nh, B, masks, seed, n_oracle):
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    batch = PS.make_config_batch(p, mu, wav, nb, B, seed, dev, masks)
    m = QFA(nb, nr, nh, dev, model_params=p); m.mu = T(mu)
    err = PS.section_errors(m, batch)
    print(f"full size npix={npix} k={nh} B={B}: one launch vs float64 sum of 512-spectrum launches")
    print("   " + "  ".join(f"{k} {v:.1e}" if isinstance(v, float) else f"{k} {v}" for k, v in err.items()))
    # run-to-run spread of the default (float atomic) accumulation: two launches of the same batch
    a1 = m.accumulate(*batch).clone(); a2 = m.accumulate(*batch).clone()
    sl = PS.sections(m)
    print("   run-to-run (atomics): " + "  ".join(
        f"{n} {float((a1[s] - a2[s]).double().norm() / a1[s].double().norm().clamp_min(1e-300)):.1e}" for n, s in sl.items()
        if n in ("accF", "sumA", "gPsi", "gOmega", "g_tau0", "g_c0", "g_beta")))
    rng = np.random.default_rng(seed)
    idx = torch.tensor(np.sort(rng.choice(B, size=n_oracle, replace=False)), device=dev)
    oe = PS.oracle_subbatch_errors(m, p, batch, idx)
    print(f"   oracle (float64) on {n_oracle} sampled spectra as their own launch: " +
          "  ".join(f"{k} {v:.1e}" for k, v in oe.items() if isinstance(v, float) and not k.endswith("_np32")))
    print("      (float32 numpy oracle vs float64, same sub-batch: " + "  ".join(f"{k[:-5]} {v:.1e}" for k, v in oe.items() if k.endswith("_np32")) + ")", flush=True)
    del batch
    torch.cuda.empty_cache()

from tests.conftest import GOLDEN
p, mu = O.load_params_npz(os.path.join(GOLDEN, "model_parameters.npz"))
wav, nb, nr = synthetic.wavelength_grid()
from qfa_amd import _lib as _L; print("lib:", _L.LIB_PATH)
case("sdss (1913, 8) B=8 (G4: red-only spectrum + dead range)", p, mu, wav, nb, 8, 20220704, red_only=(3,), dead_range=(900, 910))
case("sdss (1913, 8) B=64", p, mu, wav, nb, 64, 1)
for npix, nh, seed, B in ((2000, 8, 2, 24), (4000, 16, 3, 24), (8000, 32, 5, 12), (640, 16, 13, 24)):
    w, b_, _ = synthetic.wavelength_grid(npix)
    pp, mm = synthetic.mock_parameters(npix, b_, nh, seed=seed)
    case(f"mock ({npix}, {nh}) B={B}", pp, mm, w, b_, B, 200 + seed)
if "--full" in sys.argv:
    full(2000, 8, 10000, False, 20220702, 512)
    full(4000, 16, 100000, True, 20220703, 512)
    full(8000, 32, 2048, True, 20220705, 96)
