"""GPU box: time of one accumulate() call (pass 1 + solve + pass 2) per pass-2 form over batch sizes, N_h = 9..16.
usage: python tools/time_pass2.py [npix] [nh]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qfa_amd import QFA, _lib, synthetic
dev = torch.device("cuda:0")
npix = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 16
wav, nb, nr = synthetic.wavelength_grid(npix)
p, mu = synthetic.mock_parameters(npix, nb, nh, seed=1)
BS = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (64, 256, 500, 1000, 2000, 4000, 8000, 16000, 32000, 100000)
for B in BS:
    if B * npix > 4.2e8: continue
    d, e, z, m_, zq = synthetic.make_batch_torch(p, mu, wav, nb, B, seed=2, device=dev, return_zq=True)
    zfac = ((1.0 + zq.double()).float(), torch.tensor((wav[:nb] / synthetic.LYA).astype(np.float32), device=dev))
    out = []
    for name, fl, zf in (("f", _lib.F_PASS2_F32, None), ("x", _lib.F_PASS2_XDL, None), ("t", _lib.F_PASS2_PIXRES, None),
                         ("f+zf", _lib.F_PASS2_F32, zfac), ("x+zf", _lib.F_PASS2_XDL, zfac), ("t+zf", _lib.F_PASS2_PIXRES, zfac)):
        m = QFA(nb, nr, nh, dev, model_params=p); m.mu = torch.tensor(mu, device=dev); m.flags = fl
        for _ in range(3): m.accumulate(d, e, z if zf is None else None, m_, zfac=zf)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10 if B >= 16000 else 30
        t0.record()
        for _ in range(reps): m.accumulate(d, e, z if zf is None else None, m_, zfac=zf)
        t1.record(); torch.cuda.synchronize()
        out.append(f"{name} {t0.elapsed_time(t1) / reps:.3f}")
    print(f"npix {npix} nh {nh} B {B:6d}: " + "  ".join(out), flush=True)
