"""GPU box: is there a COHERENT difference between a spectrum's results in a launch that walks the whole pixel axis in one
accumulation chain (large batches) and in a launch that cuts it into segments (small batches: the work plan, qfa_host.h)?
Per-spectrum NLL of ONE launch over B spectra against 512-spectrum launches of the same spectra, and both against the float64
oracle on a sample.   python tools/chain_bias.py [npix] [nh] [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from qfa_amd import QFA, synthetic
    from oracle import qfa_oracle as O
    from tools import parity_sections as PS
    npix = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    nh = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    batch = PS.make_config_batch(p, mu, wav, nb, B, 20220755, dev, True)
    m = QFA(nb, nr, nh, dev, model_params=p); m.mu = torch.tensor(mu, device=dev)
    nll_big = torch.empty(B, dtype=torch.float32, device=dev)
    m.accumulate(*batch, nll=nll_big)
    _, nll_small = PS.chunked_f64(m, batch, 512)
    a, b = nll_big.double().cpu().numpy(), nll_small.double().cpu().numpy()
    r = (a - b) / np.abs(b)
    print("big vs 512-chunk launches: mean signed rel diff %.3e  rms %.3e  max %.3e" % (r.mean(), np.sqrt((r * r).mean()), np.abs(r).max()))
    idx = np.arange(0, B, max(1, B // 48))[:48]
    host = [x[torch.as_tensor(idx, device=dev)].cpu().numpy() for x in batch]
    ref = np.array([O.nll_and_grads_single(p, host[0][i], host[1][i], host[2][i], host[3][i])[0] for i in range(len(idx))])
    for name, v in (("big", a[idx]), ("chunks", b[idx])):
        e = (v - ref) / np.abs(ref)
        print("%-6s vs float64 oracle (%d spectra): mean signed %.3e  rms %.3e" % (name, len(idx), e.mean(), np.sqrt((e * e).mean())))


if __name__ == "__main__":
    main()
