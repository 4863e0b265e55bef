"""GPU box: accumulation error of ONE launch against the float64 sum of 512-spectrum launches, section by section, at growing
batch sizes (the same kernels on both sides: product arithmetic cancels, the order / width of the sums does not).
python tools/c5_sections.py [npix] [nh] [B ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, synthetic, _lib
from tools import parity_sections as PS
npix = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 32
Bs = [int(x) for x in sys.argv[3:]] or [2048, 8192, 20000]
dev = torch.device("cuda:0")
wav, nb, nr = synthetic.wavelength_grid(npix)
p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
for B in Bs:
    batch = PS.make_config_batch(p, mu, wav, nb, B, 20220755, dev, True)
    for name, fl, det in (("default", 0, False), ("det", 0, True)):
        m = QFA(nb, nr, nh, dev, model_params=p); m.mu = torch.tensor(mu, device=dev); m.flags = fl; m.deterministic = det
        e = PS.section_errors(m, batch)
        print("B %6d %-8s accF %.2e sumA %.2e gPsi %.2e gOmega %.2e" % (B, name, e["accF"], e["sumA"], e["gPsi"], e["gOmega"]), flush=True)
    del batch
    torch.cuda.empty_cache()
