"""GPU parity at BASELINE.json's FULL sizes, section by section (VERDICT r1 "weak" item 1).

For c3 (100 000 x 4000, N_h = 16, masks), c2 (10 000 x 2000, N_h = 8, no masks) and the c5 shape (8000, N_h = 32) at
the 20 000 spectra bench.py times:
  * ONE launch over the whole batch against a float64 sum of the same batch accumulated in 512-spectrum HIP launches,
    EACH section of the packed buffer on its own (accF, sumA, gPsi, gOmega, cnt, and g_tau0 / g_c0 / g_beta / sum NLL
    individually) -- the float32 atomic accumulation of sign-alternating sums over 1e5 spectra is where error grows;
  * the float64 CPU oracle's normalised gradients on a sampled sub-batch (1 024 spectra at c3) run as its own launch.
Tolerances are written next to each assert; the achieved values are recorded in profiles/r2_accuracy.txt.
"""
import numpy as np
import pytest

from tools import parity_sections as PS

pytestmark = pytest.mark.gpu

# one big launch vs float64 sum of 512-spectrum launches (achieved, profiles/r2_accuracy.txt: accF 1.3e-5 at N_h = 32,
# 4e-6 at c3; per-pixel sums <= 3e-6; scalar gradients <= 1e-5; per-spectrum NLL: 1e-6 in L2, the worst single spectrum
# of 100 000 at 1.9e-5 -- a spectrum's NLL is a difference of terms ten times its size and the two launches sum the
# pixel axis in different segmentations)
TOL_SECTION = {"accF": 3e-5, "sumA": 3e-6, "gPsi": 1e-5, "gOmega": 1e-5, "g_tau0": 3e-5, "g_c0": 3e-5,
               "g_beta": 3e-5, "sum_nll": 1e-6, "nll_per_spectrum_rel_l2": 3e-6, "nll_per_spectrum_max_rel": 5e-5}
# sampled sub-batch vs float64 oracle (achieved: loss 2e-7, per-spectrum NLL <= 2.6e-6, F <= 3.4e-5, Psi/omega <= 4.4e-6).
# The three scalar gradients of data drawn from the model itself are sums of cancelling terms (the expected gradient is
# zero; sum|terms| / |sum| = 200..4 500 on these sub-batches): a FIXED bound on the error in units of sum|terms| -- 1.5e-7 =
# 2.5 x 2^-24 (achieved 6e-8..7.5e-8, profiles/r3_accuracy.txt; the relative error of the cancelled sum is that times the
# cancellation: 1.4e-5 at c2, 3e-4 at c3, where the float32 numpy oracle itself is at 3.5e-4).
TOL_ORACLE = {"loss": 2e-6, "nll_per_spectrum_max_rel": 5e-6, "F": 1e-4, "Psi": 2e-5, "omega": 2e-5,
              "tau0_over_abs": 1.5e-7, "c0_over_abs": 1.5e-7, "beta_over_abs": 1.5e-7}


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def run_config(dev, npix, nh, B, masks, seed, n_oracle, tol_oracle=None):
    import torch
    from qfa_amd import QFA, synthetic
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    batch = PS.make_config_batch(p, mu, wav, nb, B, seed, dev, masks)
    m = QFA(nb, nr, nh, dev, model_params=p)
    m.mu = torch.tensor(mu, device=dev)
    err = PS.section_errors(m, batch)
    print("sections", npix, nh, B, err)
    assert err["finite"]
    for name in ("cnt", "n_blue", "n_spectra"):
        assert err[name] == 0.0, (name, err[name])              # integer counts below 2^24: exact
    for name, tol in TOL_SECTION.items():
        assert err[name] < tol, (name, err[name], tol)
    rng = np.random.default_rng(seed)
    idx = torch.tensor(np.sort(rng.choice(B, size=n_oracle, replace=False)), device=dev)
    oe = PS.oracle_subbatch_errors(m, p, batch, idx)
    print("oracle sub-batch", npix, nh, n_oracle, oe)
    tol_oracle = tol_oracle or TOL_ORACLE
    for k in PS.KEYS:
        assert oe["nan_pattern_" + k], k
    for name, tol in tol_oracle.items():
        assert oe[name] < tol, (name, oe[name], tol)
    del batch
    torch.cuda.empty_cache()


def test_config3_full_size_100k(dev):
    """BASELINE configs[2]: 100 000 spectra x 4000 px, N_h = 16, random pixel masks (what bench.py times)."""
    run_config(dev, 4000, 16, 100000, True, 20220703, 1024)


def test_config2_full_size_10k(dev):
    """BASELINE configs[1]: 10 000 spectra x 2000 px, N_h = 8, no masks."""
    run_config(dev, 2000, 8, 10000, False, 20220702, 1024)


def test_config5_shape_20000(dev):
    """BASELINE configs[4] shape (8000 px, N_h = 32) at the 20 000 spectra per GPU bench.py --config c5 times; oracle on 192
    of them (float64 numpy at 8000 x 32 runs ~10 spectra/s)."""
    run_config(dev, 8000, 32, 20000, True, 20220705, 192)      # (round 4: six piece products in stage 3 at N_h = 17..32 too: F 1e-4)


def test_config3_100k_one_launch_vs_float64_oracle(dev, tmp_path):
    """until round 5 the float64 oracle met the bench's kernels at 24 613 spectra at most.  Here ONE launch over
    the bench's whole c3 batch (100 000 x 4000, N_h = 16, masks; the automatic k_grads_t dispatch) against the oracle summed over the
    same 100 000 spectra on the host cores (tools/oracle_pool.py, ~40 s).  The normalised F gradient cancels 47x at this size (the data
    are drawn from the model: the expected gradient is zero): with pass 1's moments in ONE MFMA accumulator chain over the pixel axis it
    came out 1.9e-4 from the oracle (4.1e-6 of the cancelling sums); with the fresh accumulators per tile of round 5
    (qfa_xdl_kernels.h, QFA_P1_FRESH) 6.4e-5 (1.4e-6).  Both input forms."""
    import torch
    from qfa_amd import QFA, synthetic
    from tools import oracle_pool
    npix, nh, B = 4000, 16, 100000
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=20220700)
    parts = [synthetic.make_batch_torch(p, mu, wav, nb, 25000, 20220703 + 17 * i, dev, masks=True, return_zq=True) for i in range(4)]
    batch = tuple(torch.cat([q[j] for q in parts]) for j in range(4))
    zfac = ((1.0 + torch.cat([q[4] for q in parts])).contiguous(), torch.tensor((wav[:nb] / synthetic.LYA).astype(np.float32), device=dev))
    del parts
    host = {k: x.cpu().numpy() for k, x in zip(("delta", "error", "zabs", "mask"), batch)}
    ol, og, sums, counts = oracle_pool.oracle_sums(p, host, str(tmp_path / "oracle"))
    del host
    for name, zf in (("zabs", None), ("factored", zfac)):
        m = QFA(nb, nr, nh, dev, model_params=p)
        m.mu = torch.tensor(mu, device=dev)
        acc = m.accumulate(batch[0], batch[1], batch[2] if zf is None else None, batch[3], zfac=zf).clone()
        loss, g = m._finalize(acc, True)
        e = {}
        for k in ("F", "Psi", "omega"):
            ref = np.asarray(og[k], dtype=np.float64)
            ok = ~np.isnan(ref)
            e[k] = float(np.linalg.norm(g[k].cpu().numpy().astype(np.float64)[ok] - ref[ok]) / np.linalg.norm(ref[ok]))
        e["loss"] = abs(loss.item() - ol) / abs(ol)
        print("100 000 x 4000, N_h = 16, one launch vs float64 oracle:", name, e)
        # achieved: F 6.4e-5, Psi 6.3e-6, omega 1.3e-5, loss 1.9e-7 (unchanged by the float16 stages of pass 2; 6.9e-5 with pass 1 on
        # float16 pieces, the option that is off).  The tolerance of the table is 1e-4 for F; the bar here sits where a change of the
        # arithmetic shows (deterministic at this size)
        assert e["F"] < 8e-5 and e["Psi"] < 1e-5 and e["omega"] < 2e-5 and e["loss"] < 6e-7, (name, e)
    del batch
    torch.cuda.empty_cache()
