"""Diagnostic (GPU box): run-to-run spread of the packed buffer and the per-spectrum NLL, default (atomic) and
deterministic (slab) accumulation, XDL and f32 forms of pass 2.  usage: diag_runtorun.py npix nh B [B ...]"""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qfa_amd import QFA, synthetic
from tools import parity_sections as PS

dev = torch.device("cuda:0")
npix, nh = int(sys.argv[1]), int(sys.argv[2])
wav, nb, nr = synthetic.wavelength_grid(npix)
p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
for B in [int(x) for x in sys.argv[3:]]:
    batch = PS.make_config_batch(p, mu, wav, nb, B, 20220703, dev, True)
    m = QFA(nb, nr, nh, dev, model_params=p)
    sl = PS.sections(m)
    for form in os.environ.get("DIAG_FORMS", "xdl,f32").split(","):
        m.flags = 0x1 if form == "f32" else 0x2          # _lib.F_PASS2_F32 / F_PASS2_XDL
        m.deterministic = True
        NR = int(os.environ.get('DIAG_RUNS', '4'))
        nll = [torch.empty(B, device=dev) for _ in range(NR)]
        runs = [m.accumulate(*batch, nll=nll[i]).clone() for i in range(NR)]
        torch.cuda.synchronize()
        nd = [int((nll[0] != nll[i]).sum()) for i in range(1, NR)]
        print(f"== B {B} form {form} deterministic: spectra whose NLL differs from run 0: {nd}")
        for n in ("accF", "sumA", "gPsi", "gOmega", "g_tau0"):
            s = sl[n]
            dif = [(runs[0][s] != runs[i][s]) for i in range(1, NR)]
            cnt = [int(d.sum()) for d in dif]
            where = ""
            if n in ("gPsi", "sumA") and max(cnt) > 0:
                idx = sorted(set(torch.nonzero(torch.stack(dif).any(0)).flatten().tolist()))
                where = f" px(all runs) first-of-tile {sorted(set((i // 32, i % 2) for i in idx))}"
            if n == "accF" and max(cnt) > 0:
                i = int(np.argmax(cnt))
                idx = torch.unique(torch.nonzero(dif[i]).flatten() // nh).tolist()
                where = f" px {idx[:24]} ({len(idx)} pixels)"
            print(f"   {n:8s} elements differing {cnt}  rel {[f'{rel(runs[0][s], runs[i][s]):.1e}' for i in range(1, NR)]}{where}", flush=True)
    del batch
