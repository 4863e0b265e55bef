"""Run the hot path once on seeded synthetic input and dump every output to an .npz (GPU box only).

    python tools/dump_hot_path.py OUT.npz NPIX NH B [deterministic] [zfac] [pixres] [lib=PATH]

Used by tests/test_tracked_loads.py to compare two builds of the library (lib=PATH selects the one this process
loads) bit for bit: the packed accumulation buffer, the per-spectrum NLL and the five prediction outputs.
"""
import sys
import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from qfa_amd import QFA, synthetic, _lib   # noqa: E402
for a in sys.argv[5:]:
    if a.startswith("lib="):
        _lib.LIB_PATH = a[4:]

out, npix, nh, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
det = "deterministic" in sys.argv[5:]
zf = "zfac" in sys.argv[5:]               # the factored-z input form (its kernels have their own request counts)
dev = torch.device("cuda:0")
wav, nb, nr = synthetic.wavelength_grid(npix)
p, mu = synthetic.mock_parameters(npix, nb, nh, seed=4242)
b = synthetic.make_batch_numpy(p, mu, wav, nb, B, seed=4243)
T = lambda x: torch.tensor(x, device=dev)
m = QFA(nb, nr, nh, dev, model_params=p)
m.mu = T(mu)
m.deterministic = det
if "pixres" in sys.argv[5:]:              # N_h = 9..16: the pixel-resident form of pass 2 (k_grads_t)
    m.flags = _lib.F_PASS2_PIXRES
nll = torch.empty(B, device=dev)
zfac = (T(1.0 + b["zqso"].astype(np.float64)).float(), T((wav[:nb] / synthetic.LYA).astype(np.float32))) if zf else None
acc = m.accumulate(T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"]), nll=nll, zfac=zfac).clone()
pred = m.predict(T(b["flux"]), T(b["error"]), T(b["zabs"]), T(b["mask"]), zfac=zfac)
np.savez(out, acc=acc.cpu().numpy(), nll=nll.cpu().numpy(), **{f"pred{i}": x.cpu().numpy() for i, x in enumerate(pred)})
