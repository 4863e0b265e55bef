"""Golden fixture g13 for the reference's second shipped model, data/model_parameters_desi.npz (N_pix = 9243,
N_b = 2238, N_h = 8), by running the *imported reference* (CPU, dense O(N_pix^3): minutes) in the build container:

    python tests/golden/make_golden_desi.py

Writes model_parameters_desi.npz (a copy of the reference's data file: MIT, see ATTRIBUTION.md) and g13_desi.npz:
inputs = seed + the wavelength grid recipe below; outputs = prediction_for_single_spectra on one seeded mock spectrum
(full mask and with the blue side masked) and loglikelihood_and_gradient_for_single_spectra on another.  The file does
not say which rest-frame grid the DESI model was trained on; the reference's functions take zabs / masks directly, so
any grid with 2238 blue pixels exercises the same code -- linear, 1040 A + 0.07851 A per pixel.
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, REPO, import_reference  # noqa: E402


def desi_grid():
    wav = 1040.0 + (1215.67 - 1040.0) / 2237.5 * np.arange(9243)
    nb = int(np.sum(wav < 1215.67))
    return wav, nb, len(wav) - nb


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, optimizer, utils = import_reference()
    from qfa_amd import synthetic
    dev = torch.device("cpu")
    wav, nb, nr = desi_grid()
    assert (len(wav), nb) == (9243, 2238)
    src = os.path.join(REF, "data", "model_parameters_desi.npz")
    dst = os.path.join(HERE, "model_parameters_desi.npz")
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)

    def T(x, dt=torch.float32):
        return torch.tensor(np.asarray(x), dtype=dt)

    m = model.QFA(nb, nr, 8, dev)
    m.load_from_npz(src)
    shipped = {k: getattr(m, k).numpy().copy() for k in ("F", "Psi", "omega", "tau0", "c0", "beta")}
    mu = m.mu.numpy().copy()
    seed = 20220713
    b = synthetic.make_batch_numpy(shipped, mu, wav, nb, 2, seed=seed)
    out = {"seed": seed}
    for tag, mk in (("full", b["mask"][0]), ("red", b["mask"][0] & (np.arange(len(wav)) >= nb))):
        ll, hm, hc, cont, unc = m.prediction_for_single_spectra(T(b["flux"][0]), T(b["error"][0]), T(b["zabs"][0]),
                                                                T(mk, torch.bool))
        out.update({f"ll_{tag}": ll.numpy(), f"hmean_{tag}": hm.numpy(), f"hcov_{tag}": hc.numpy(),
                    f"cont_{tag}": cont.numpy(), f"unc_{tag}": unc.numpy()})
        print("predict", tag, float(ll), flush=True)
    nll, g = m.loglikelihood_and_gradient_for_single_spectra(T(b["delta"][1]), T(b["error"][1]), T(b["zabs"][1]),
                                                             T(b["mask"][1], torch.bool))
    out["nll"] = nll.numpy()
    out.update({f"g_{k}": v.detach().numpy() for k, v in g.items()})
    print("nll", float(nll), flush=True)
    np.savez_compressed(os.path.join(HERE, "g13_desi.npz"), **out)


if __name__ == "__main__":
    main()
