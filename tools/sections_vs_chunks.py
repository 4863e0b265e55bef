"""One launch against the float64 sum of chunked launches, section by section of the packed buffer (diagnostic):
python tools/sections_vs_chunks.py B CHUNK [flags]"""
import os, sys
if __name__ == "__main__":
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, REPO)
    import numpy as np, torch
    from qfa_amd import QFA, synthetic
    import qfa_amd.model as M
    from tools import parity_sections as PS
    M.AUTO_FACTOR_ZABS = False
    B, chunk = int(sys.argv[1]), int(sys.argv[2])
    flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
    npix, nh = 4000, 16
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    batch = PS.make_config_batch(p, mu, wav, nb, B, 20220703, dev, True)
    m = QFA(nb, nr, nh, dev, model_params=p); m.mu = torch.tensor(mu, device=dev); m.flags = flags
    e = PS.section_errors(m, batch, chunk)
    print(B, chunk, hex(flags), {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in e.items()})
