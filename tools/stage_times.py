"""GPU box: stage times (image, pass 1, solve, pass 2) of accumulate() at given shapes: python tools/stage_times.py NH B NPIX..."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, synthetic
nh, B = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
for npix in (int(x) for x in sys.argv[3:]):
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=1)
    d, e, z, m_ = synthetic.make_batch_torch(p, mu, wav, nb, B, 7, dev, masks=True)
    m = QFA(nb, nr, nh, dev, model_params=p); m.mu = torch.tensor(mu, device=dev)
    rec = []
    for it in range(10):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        for ev in evs: ev.record()
        m.accumulate(d, e, z, m_, events=evs)
        torch.cuda.synchronize()
        if it >= 4: rec.append([evs[i].elapsed_time(evs[i + 1]) for i in range(4)])
    r = np.median(np.array(rec), axis=0)
    print("N_pix %5d (nb %4d) N_h %2d B %6d: image %.3f  pass 1 %.3f  solve %.3f  pass 2 %.3f  | per 1e6 pixel-spectra: p1 %.4f p2 %.4f" % (
        npix, nb, nh, B, r[0], r[1], r[2], r[3], r[1] / (B * npix / 1e6), r[3] / (B * npix / 1e6)), flush=True)
    del d, e, z, m_, m
    torch.cuda.empty_cache()
