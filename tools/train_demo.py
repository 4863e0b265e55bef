"""End-to-end demo on synthetic spectra drawn from a known model: train from the reference's random initialisation
and report the epoch losses and the continuum error of the fitted model on held-out spectra (GPU box)."""
import os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, Adam, step_scheduler, synthetic
from qfa_amd.dataloader import DeviceDataloader

dev = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(0)
npix, nh, N = 1200, 8, 20000
wav, nb, nr = synthetic.wavelength_grid(npix)
p_true, mu = synthetic.mock_parameters(npix, nb, nh, seed=11)
b = synthetic.make_batch_numpy(p_true, mu, wav, nb, N + 256, seed=12, masks=True)
tr = {k: v[:N] for k, v in b.items()}
dl = DeviceDataloader(tr["flux"], tr["error"], tr["zqso"], wav, batch_size=500, device=dev, shuffle=True)
model = QFA(nb, nr, nh, dev)
model.random_init_func()
opt = Adam(model.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-2, weight_decay=1e-3)
t0 = time.time()
with tempfile.TemporaryDirectory() as d:
    nep = int(os.environ.get("QFA_DEMO_EPOCHS", "40"))
    model.train(opt, dl, nep, d, quiet=nep > 60, use_graph="--graph" in sys.argv)
torch.cuda.synchronize()
print(f"{nep} epochs x {N} spectra x {npix} px, N_h={nh}, batch 500: {time.time() - t0:.1f} s wall")
T = lambda x: torch.tensor(x[N:], device=dev)
for name, m in (("fitted", model), ("true parameters", QFA(nb, nr, nh, dev, model_params=p_true))):
    m.mu = model.mu
    ll, hm, hc, cont, unc = m.predict(T(b["flux"]), T(b["error"]), T(b["zabs"]), T(b["mask"]))
    truth = torch.tensor(b["continuum"][N:], device=dev) if "continuum" in b else None
    print(name, "held-out mean -loglik per spectrum:", float(ll.mean()),
          "" if truth is None else f"continuum rms rel err {float(((cont - truth) / truth).pow(2).mean().sqrt()):.4f}")
