"""GPU box: where does the F-gradient error at N_h = 17..32 come from?  c5's shape at B spectra against the float64 oracle pool:
python tools/c5_probe.py [B] [npix] [nh]"""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qfa_amd import QFA, synthetic, _lib
from tools import oracle_pool, parity_sections as PS

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    npix = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    nh = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    batch = PS.make_config_batch(p, mu, wav, nb, B, 20220755, dev, True)
    host = {k: x.cpu().numpy() for k, x in zip(("delta", "error", "zabs", "mask"), batch)}
    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        ol, og, sums, counts = oracle_pool.oracle_sums(p, host, td)
    print("oracle %.1f s for %d spectra" % (time.time() - t0, B), flush=True)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
    sl = None
    for name, fl, det in (("six", 0, False), ("six det", 0, True), ("fast", _lib.F_S3_FAST, False), ("fast det", _lib.F_S3_FAST, True)):
        m = QFA(nb, nr, nh, dev, model_params=p); m.mu = torch.tensor(mu, device=dev); m.flags = fl; m.deterministic = det
        acc = m.accumulate(*batch).clone()
        loss, g = m._finalize(acc, True)
        sl = PS.sections(m)
        accF = acc[sl["accF"]].double().cpu().numpy().reshape(npix, nh)
        sumA = acc[sl["sumA"]].double().cpu().numpy()
        F = np.asarray(p["F"], np.float64)
        # the two cancelling terms of gF = (F sumA - accF) / cnt, against the oracle's raw sums
        terms = np.linalg.norm(F * sumA[:, None]) + np.linalg.norm(accF)
        gF_sum = F * sumA[:, None] - accF
        print("%-9s F %.3e  F_err/terms %.3e  cancellation %.1f  Psi %.2e omega %.2e loss %.1e | raw gF sum vs oracle %.3e" % (
            name, rel(g["F"].cpu().numpy(), og["F"]), np.linalg.norm(g["F"].cpu().numpy().astype(np.float64) * counts["F"] - sums["F"]) / terms,
            terms / np.linalg.norm(sums["F"]), rel(g["Psi"].cpu().numpy(), og["Psi"]), rel(g["omega"].cpu().numpy(), og["omega"]),
            abs(loss.item() - ol) / abs(ol), rel(gF_sum, sums["F"])), flush=True)


if __name__ == "__main__":      # (the oracle pool spawns workers that re-import this module: they must not touch the GPU)
    main()
