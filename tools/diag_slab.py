"""Diagnostic (GPU box): in deterministic mode the slab holds one row of tile partials per block of 64 spectra.
Compare the slabs of repeated runs: which (block, tile) pairs differ, and where that tile sits in the block's order."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from qfa_amd import QFA, synthetic
from tools import parity_sections as PS
dev = torch.device("cuda:0")
npix, nh, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wav, nb, nr = synthetic.wavelength_grid(npix)
p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
batch = PS.make_config_batch(p, mu, wav, nb, B, 20220703, dev, True)
m = QFA(nb, nr, nh, dev, model_params=p)
m.deterministic = True
NF = npix * nh + 3 * npix + nb
nblk = (B + 63) // 64
ntiles = (npix + 31) // 32
slabs = []
for i in range(int(os.environ.get("DIAG_RUNS", "6"))):
    m.accumulate(*batch)
    torch.cuda.synchronize()
    slabs.append(m._ws["slab"][: nblk * NF * 4].view(torch.float32).view(nblk, NF).clone())
for i in range(1, len(slabs)):
    d = slabs[0] != slabs[i]
    blks = torch.nonzero(d.any(1)).flatten().tolist()
    out = []
    for b in blks[:12]:
        cols = torch.nonzero(d[b]).flatten()
        sumA = cols[(cols >= npix * nh) & (cols < npix * nh + npix)] - npix * nh
        tiles = sorted(set((sumA // 32).tolist()))
        accf_px = sorted(set((cols[cols < npix * nh] // nh // 32).tolist()))
        rot = (b * 2654435761 % 2**32) % min(ntiles, 32)
        odd = sorted(set((sumA % 2).tolist()))
        out.append(f"blk {b} rot {rot} tiles(sumA) {tiles} parity {odd} tiles(accF) {accf_px}")
    print(f"run {i}: {len(blks)} blocks differ;", " | ".join(out), flush=True)
# values of the first differing block/tile
d = slabs[0] != slabs[1]
blks = torch.nonzero(d.any(1)).flatten().tolist()
if blks:
    b = blks[0]
    cols = torch.nonzero(d[b]).flatten()
    sumA = cols[(cols >= npix * nh) & (cols < npix * nh + npix)] - npix * nh
    t = int(sumA[0]) // 32
    for name, off in (("sumA", npix * nh), ("gPsi", npix * nh + npix), ("gOm", npix * nh + 2 * npix), ("cnt", npix * nh + 2 * npix + nb)):
        s = slice(off + 32 * t, off + 32 * t + 32)
        print(name, "run0", [f"{v:.6g}" for v in slabs[0][b, s].tolist()])
        print(name, "run1", [f"{v:.6g}" for v in slabs[1][b, s].tolist()])
    px = 32 * t + 1
    print("accF px", px, "run0", [f"{v:.6g}" for v in slabs[0][b, px * nh:(px + 1) * nh].tolist()])
    print("accF px", px, "run1", [f"{v:.6g}" for v in slabs[1][b, px * nh:(px + 1) * nh].tolist()])
