"""GPU box: the XDL posterior writers (k_predict_x: plain, two groups per wave, re-aligned stores; k_predict_x32) against the float32-MFMA
writer k_predict_out on random shapes, every element of cont / unc.  usage: python tools/writer_sweep.py [seed] [cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, _lib, synthetic
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(ncase):
    nh = int(rng.choice([1, 3, 5, 8, 8, 8, 9, 12, 16, 16, 20, 32]))
    npix = int(rng.choice([33, 64, 97, 160, 333, 640, 1000, 1913, 1920, 2000, 3111])) if rng.integers(0, 2) else int(rng.integers(33, 2500))
    B = int(rng.choice([1, 3, 16, 17, 63, 64, 65, 127, 128, 129, 200, 513, 1500, 4100]))
    if nh > 16: npix = min(npix, 1200); B = min(B, 600)
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=case)
    d, e, z, m_ = synthetic.make_batch_torch(p, mu, wav, nb, B, 100 + case, dev, masks=bool(rng.integers(0, 2)))
    m = QFA(nb, nr, nh, dev, model_params=p); m.mu = torch.tensor(mu, device=dev)
    off = int(rng.choice([0, 0, 1, 5, 16, 31]))
    n = B * npix
    big = torch.full((2 * n + 512,), -3.0, dtype=torch.float32, device=dev)
    cv, uv = big[off: off + n].view(B, npix), big[n + 128 + off: n + 128 + off + n].view(B, npix)
    out = (torch.empty((B,), device=dev), torch.empty((B, nh), device=dev), torch.empty((B, nh, nh), device=dev), cv, uv)
    m.predict(d, e, z, m_, out=out)
    m.flags = _lib.F_PREDICT_F32
    ll, hm, hc, c2, u2 = m.predict(d, e, z, m_)
    torch.cuda.synchronize()
    ec = float((cv - c2).abs().max() / c2.abs().max()); eu = float((uv - u2).abs().max() / u2.abs().max())
    guard = torch.ones_like(big, dtype=torch.bool); guard[off: off + n] = False; guard[n + 128 + off: n + 128 + off + n] = False
    clean = bool((big[guard] == -3.0).all())
    ok = ec <= 3e-6 and eu <= 6e-6 and clean and bool(torch.isfinite(cv).all()) and bool(torch.isfinite(uv).all())
    print(f"case {case:3d} npix {npix:5d} nh {nh:2d} B {B:5d} off {off:2d}: cont {ec:.1e} unc {eu:.1e} guard {'ok' if clean else 'TOUCHED'} {'ok' if ok else 'FAIL'}", flush=True)
    bad += not ok
print("cases with a failure:", bad)
sys.exit(1 if bad else 0)
