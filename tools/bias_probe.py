"""GPU box: SIGNED bias of the per-pixel sums (sumA = sum_s wD A^3, gPsi = sum_s A^2 dG, gOmega = sum_s dG zd) of one HIP
launch against the float64 oracle, normalised by the sum of |terms| -- a coherent bias of a fraction of an ulp is invisible
in relative L2 but adds up in the strongly cancelling scalar gradients."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import qfa_oracle as O
from qfa_amd import QFA, synthetic
from tools import parity_sections as PS
from tests.conftest import GOLDEN
dev = torch.device("cuda:0")
T = lambda x: torch.tensor(x, device=dev)
def run(name, p, mu, wav, nb, B, seed, flags=0):
    b = synthetic.make_batch_numpy(p, mu, wav, nb, B, seed=seed)
    m = QFA(nb, len(wav) - nb, p["F"].shape[1], dev, model_params=p); m.mu = T(mu); m.flags = flags
    acc = m.accumulate(T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"])).double().cpu().numpy()
    sl = PS.sections(m)
    pp = O._as_params(p, np.float64)
    npix = len(wav)
    ref = {k: np.zeros(npix) for k in ("sumA", "gPsi", "gOmega", "dS", "u2")}
    ab = {k: np.zeros(npix) for k in ref}
    for s in range(B):
        w = b["mask"][s]
        A, zdep, D = O.pixel_terms(p, b["error"][s], b["zabs"][s])
        wD, d, M, C, y, u, nll = O._lowrank_core(pp["F"], A, D, w, b["delta"][s].astype(np.float64))
        q = np.einsum("ia,ab,ib->i", pp["F"], np.linalg.inv(C), pp["F"])
        dS = wD - (wD * A) ** 2 * q
        dG = .5 * (dS - u * u)
        t = {"sumA": wD * A ** 3, "gPsi": A * A * dG, "gOmega": dG * zdep, "dS": .5 * A * A * dS, "u2": .5 * A * A * u * u}
        for k in t:
            ref[k] += t[k]; ab[k] += np.abs(t[k])
    print(name, "flags", flags)
    for k in ("sumA", "gPsi", "gOmega"):
        h = acc[sl[k]]; r = ref[k][:len(h)]; a = ab[k][:len(h)]
        blue = slice(0, nb)
        print(f"   {k}: rel-L2 {np.linalg.norm(h-r)/np.linalg.norm(r):.1e}  signed bias sum(h-r)/sum|terms| all {np.sum(h-r)/np.sum(a):+.2e} blue {np.sum((h-r)[blue])/np.sum(a[blue]):+.2e}"
              + (f" red {np.sum((h-r)[nb:])/np.sum(a[nb:]):+.2e}" if len(h) > nb else ""))
    print(f"   (gPsi = dS part - u2 part; sum|dS part| {np.sum(ab['dS']):.3e} sum|u2 part| {np.sum(ab['u2']):.3e} sum|A2 dG| {np.sum(ab['gPsi']):.3e})", flush=True)
p, mu = O.load_params_npz(os.path.join(GOLDEN, "model_parameters.npz"))
wav, nb, nr = synthetic.wavelength_grid()
run("G5-like (64 spectra, shipped parameters)", p, mu, wav, nb, 64, 20220705, 0)
run("G5-like (64 spectra, shipped parameters)", p, mu, wav, nb, 64, 20220705, 2)
w, b_, _ = synthetic.wavelength_grid(4000)
pp, mm = synthetic.mock_parameters(4000, b_, 16, seed=3)
run("mock (4000, 16) B=24", pp, mm, w, b_, 24, 203)
