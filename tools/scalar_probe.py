"""GPU box: errors of the three scalar gradients (sums of strongly cancelling per-pixel terms) against the float64 oracle
on the golden batches G4 / G5 and two mock shapes, next to the reference's own float32 result (golden files) and the
float32 numpy oracle; also error / sum|terms| (the condition-number-free figure).  run through tools/with_lib.py for a library variant."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import qfa_oracle as O
from qfa_amd import QFA, synthetic
from tests.conftest import GOLDEN
dev = torch.device("cuda:0")
T = lambda x: torch.tensor(x, device=dev)
def absterms(p, b, nb):
    tot = np.zeros(3)
    pp = O._as_params(p, np.float64)
    for s in range(len(b["delta"])):
        w = b["mask"][s]
        A, zdep, D = O.pixel_terms(p, b["error"][s], b["zabs"][s])
        wD, d, M, C, y, u, nll = O._lowrank_core(pp["F"], A, D, w, b["delta"][s].astype(np.float64))
        q = np.einsum("ia,ab,ib->i", pp["F"], np.linalg.inv(C), pp["F"])
        dG = .5 * (wD - (wD * A) ** 2 * q - u * u)
        z = b["zabs"][s].astype(np.float64); pw = (1 + z) ** pp["beta"]; root = 1 - pp["tau0"] * pw - pp["c0"]
        e = np.where(w[:nb], dG[:nb] * (pp["omega"] * zdep[:nb]) * zdep[:nb] * 2 * root, 0)
        tot += [np.abs(e * pw).sum(), np.abs(e).sum(), np.abs(e * pp["tau0"] * pw * np.log(1 + z)).sum()]
    return tot
def run(name, p, mu, wav, nb, B, seed, gold=None, flags=0, **kw):
    b = synthetic.make_batch_numpy(p, mu, wav, nb, B, seed=seed, **kw)
    m = QFA(nb, len(wav) - nb, p["F"].shape[1], dev, model_params=p); m.mu = T(mu); m.flags = flags
    loss, g, sums, counts = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"], return_sums=True)
    l32, g32 = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"], dtype=np.float32)
    _, gh = m.forward(T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"]))
    at = absterms(p, b, nb)
    out = []
    for i, k in enumerate(("tau0", "c0", "beta")):
        o = float(g[k]); h = float(gh[k].item()); cnt = float(counts[k])
        line = f"{k}: hip {abs(h-o)/abs(o):.1e} np32 {abs(float(g32[k])-o)/abs(o):.1e}"
        if gold is not None:
            line += f" reference-f32 {abs(float(gold['g_'+k])-o)/abs(o):.1e}"
        line += f" | hip err/sum|terms| {abs(h-o)*cnt/at[i]:.1e} (cancellation {at[i]/abs(o*cnt):.0f}x)"
        out.append(line)
    print(name + "\n   " + "\n   ".join(out), flush=True)
p, mu = O.load_params_npz(os.path.join(GOLDEN, "model_parameters.npz"))
wav, nb, nr = synthetic.wavelength_grid()
from qfa_amd import _lib as _L; print("lib:", _L.LIB_PATH)
g4 = np.load(os.path.join(GOLDEN, "g4_forward.npz")); g5 = np.load(os.path.join(GOLDEN, "g5_step.npz"))
for fl in (0, 2):
    print("flags", fl)
    run("G4 (8 spectra, shipped parameters)", p, mu, wav, nb, 8, int(g4["seed"]), g4, fl, red_only=(3,), dead_range=(900, 910))
    run("G5 (128 spectra, shipped parameters)", p, mu, wav, nb, 128, int(g5["seed"]), g5, fl)
for npix, nh, seed, B in ((2000, 8, 2, 24), (4000, 16, 3, 24)):
    w, b_, _ = synthetic.wavelength_grid(npix)
    pp, mm = synthetic.mock_parameters(npix, b_, nh, seed=seed)
    run(f"mock ({npix}, {nh}) B={B}", pp, mm, w, b_, B, 200 + seed)
