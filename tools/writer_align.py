"""GPU box: posterior-writer time against the row length N_pix (row alignment of cont / unc): python tools/writer_align.py NH B NPIX..."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, synthetic
nh, B = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
for npix in (int(x) for x in sys.argv[3:]):
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=1)
    d, e, z, m = synthetic.make_batch_torch(p, mu, wav, nb, B, 7, dev, masks=True)
    model = QFA(nb, nr, nh, dev, model_params=p); model.mu = torch.tensor(mu, device=dev)
    ts = []
    out = None
    for it in range(8):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for ev in evs: ev.record()
        out = model.predict(d, e, z, m, events=evs, out=out)
        torch.cuda.synchronize()
        if it >= 3: ts.append(evs[2].elapsed_time(evs[3]))
    lo = min(ts)
    print("N_pix %5d (row %6d B, mod 128 = %3d, mod 8 = %d)  N_h %2d  B %6d: writer min %.3f median %.3f ms = %.0f GB/s" % (
        npix, 4 * npix, (4 * npix) % 128, (4 * npix) % 8, nh, B, lo, float(np.median(ts)), 8e-6 * B * npix / lo), flush=True)
    del d, e, z, m, out, model
    torch.cuda.empty_cache()
