"""Diagnostic build only (QFA_GX_STAMPS): per-tile cycle anatomy of k_grads_x, role A (wave 0) and role B (wave 4) of block 300."""
import ctypes as C, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qfa_amd import QFA, _lib, synthetic
from tools import parity_sections as PS
dev = torch.device("cuda:0")
npix, nh, B = 4000, 16, 100000
wav, nb, nr = synthetic.wavelength_grid(npix)
p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
batch = PS.make_config_batch(p, mu, wav, nb, B, 20220703, dev, True)
m = QFA(nb, nr, nh, dev, model_params=p)
for _ in range(4):
    m.accumulate(*batch)
torch.cuda.synchronize()
out = (C.c_ulonglong * 32)()
assert _lib.lib().qfa_gx_debug_stamps(out) == 0
s = np.array(list(out), dtype=np.float64)
for name, o in (("role A blue", 0), ("role A red", 8), ("role B", 16)):
    n = max(s[o + 4], 1)
    print(f"{name}: tiles {int(s[o+4])}  issue(DMA+loads) {s[o]/n:.0f}  compute {s[o+1]/n:.0f}  flush {s[o+2]/n:.0f}  barrier {s[o+3]/n:.0f}  total {(s[o]+s[o+1]+s[o+2]+s[o+3])/n:.0f} cycles/tile")
