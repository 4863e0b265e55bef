"""Step latency of small batches, eager launches vs the captured hipGraph (GPU box).
usage: python tools/step_latency.py [B ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, Adam, step_scheduler, synthetic
from qfa_amd.model import StepGraph

dev = torch.device("cuda:0")
wav, nb, nr = synthetic.wavelength_grid(None)          # the shipped model's grid: 1913 pixels
npix = len(wav)
for B in [int(x) for x in sys.argv[1:]] or [128, 500, 2000]:
    p, mu = synthetic.mock_parameters(npix, nb, 8, seed=1)
    batch = synthetic.make_batch_torch(p, mu, wav, nb, B, 7, dev, masks=True)
    res = {}
    for mode in ("eager", "graph"):
        m = QFA(nb, nr, 8, dev, model_params=p)
        opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)
        sg = StepGraph(m, opt, B)
        for d, s in zip(sg.buf, batch):
            d.copy_(s)
        step = (lambda: m.step(opt, *batch)) if mode == "eager" else sg.run
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / n * 1e6
    print(f"B={B:5d} x {npix} px, N_h=8: eager {res['eager']:.1f} us/step, graph {res['graph']:.1f} us/step "
          f"({B / res['graph'] * 1e6:.3g} spectra/s)")
