"""GPU box: time of the posterior writer against the byte offset between its two output arrays (cont, unc), both carved
from one allocation.  usage: python tools/writer_skew.py NPIX NH B"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, synthetic
npix, nh, B = (int(x) for x in sys.argv[1:4])
dev = torch.device("cuda:0")
wav, nb, nr = synthetic.wavelength_grid(npix)
p, mu = synthetic.mock_parameters(npix, nb, nh, seed=1)
d, e, z, m = synthetic.make_batch_torch(p, mu, wav, nb, B, 7, dev, masks=True)
model = QFA(nb, nr, nh, dev, model_params=p); model.mu = torch.tensor(mu, device=dev)
f32 = torch.float32
ll = torch.empty((B,), dtype=f32, device=dev); hm = torch.empty((B, nh), dtype=f32, device=dev); hc = torch.empty((B, nh, nh), dtype=f32, device=dev)
n = B * npix
big = torch.empty(2 * n + (64 << 20) // 4, dtype=f32, device=dev)
print("base address %x, array bytes %d (mod 4096: %d, mod 2 MiB: %d)" % (big.data_ptr(), 4 * n, (4 * n) % 4096, (4 * n) % (2 << 20)))
def run(skew_bytes, first=0):
    cont = big[first // 4: first // 4 + n].view(B, npix)
    o = first // 4 + n + skew_bytes // 4
    unc = big[o: o + n].view(B, npix)
    out = (ll, hm, hc, cont, unc)
    ts = []
    for it in range(6):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for ev in evs: ev.record()
        model.predict(d, e, z, m, events=evs, out=out)
        torch.cuda.synchronize()
        if it >= 2: ts.append(evs[2].elapsed_time(evs[3]))
    return min(ts), float(np.median(ts))
for skew in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 8192, 4 << 20, 32 << 20):
    lo, med = run(skew)
    print("unc = cont + array + %9d B (offset mod 4096 %5d, mod 64Ki %6d, mod 2Mi %8d): writer min %.3f median %.3f ms  (%.0f GB/s)" % (
        skew, (4 * n + skew) % 4096, (4 * n + skew) % 65536, (4 * n + skew) % (2 << 20), lo, med, 8e-6 * n / lo))
for first in (0, 128, 1024, 4096, 65536):
    lo, med = run(0, first)
    print("cont at base + %6d: writer min %.3f median %.3f ms" % (first, lo, med))
