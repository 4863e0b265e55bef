import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, _lib, synthetic
dev = torch.device("cuda:0")
npix, nh = 4000, 16
wav, nb, nr = synthetic.wavelength_grid(npix)
p, mu = synthetic.mock_parameters(npix, nb, nh, seed=1)
for B in (1500, 1900, 2000, 2048, 2100, 3000, 4000, 4100, 6000):
    d, e, z, m_, zq = synthetic.make_batch_torch(p, mu, wav, nb, B, seed=2, device=dev, return_zq=True)
    zfac = ((1.0 + zq.double()).float(), torch.tensor((wav[:nb] / synthetic.LYA).astype(np.float32), device=dev))
    m = QFA(nb, nr, nh, dev, model_params=p); m.mu = torch.tensor(mu, device=dev); m.flags = _lib.F_PASS2_XDL | _lib.F_PASS2_PIXRES
    ts = []
    for rep in range(6):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); m.accumulate(d, e, None, m_, zfac=zfac); t1.record(); torch.cuda.synchronize()
        ts.append(round(t0.elapsed_time(t1), 3))
    print(B, ts, flush=True)
