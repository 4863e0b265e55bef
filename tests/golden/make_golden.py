"""Generate golden fixtures by running the *imported reference* (CPU) in the build container.

Run once, here only:   python tests/golden/make_golden.py
The reference lives read-only at /root/reference and never travels: only inputs (seeds or
small arrays) and the reference's OUTPUTS are written, as .npz files next to this script.
The tests regenerate seeded inputs with ``qfa_amd.synthetic`` and compare against these.

Import recipe: stub ``yacs.config.CfgNode`` (yacs is not installed),
cwd = /root/reference/QFA because QFA/utils.py:144 opens ./Lyman_series.csv relative to cwd.

Fixtures written (SURVEY.md 8(c) list):
  g1_g2_predict.npz   prediction_for_single_spectra on the shipped SDSS spectrum, full mask
                      and blue side masked; also carries the file's own ll/h/our/ll_red/...
  g3_single.npz       loglikelihood_and_gradient_for_single_spectra on 4 seeded mock spectra
  g4_forward.npz      forward on B=8 with an all-masked pixel range and a red-only spectrum
  g5_step.npz         forward -> Adam.update -> clip on B=128 (config 1)
  g6_smooth_clip.npz  smooth() and clip() on the shipped parameters
  g7_adam.npz         Adam trace, 3 epochs x 2 batches, step_scheduler(0.9, 2)
  g8_woodbury.npz     MatrixInverse / MatrixLogDet at n=64, k=4
  g9_tau.npz          tau(which, series), tauHI, omega_func on a z grid
  g10_k16.npz         k=16 case: reference float32 loss (inf) next to per-spectrum values
  g11_dataprep.npz    tau_total (1 and 2 Lyman series), zabs, mu estimate + smooth, delta (dataloader call sites)
  sdss_spectrum.npz   inputs flux/error/z of data/spec-4321-55504-0114.npz (MIT, see ATTRIBUTION)
  model_parameters.npz  copy of data/model_parameters.npz (MIT, see ATTRIBUTION)
"""
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def import_reference():
    yacs = types.ModuleType("yacs")
    cfg = types.ModuleType("yacs.config")

    class CfgNode(dict):
        def __getattr__(self, k):
            return self[k]

        def __setattr__(self, k, v):
            self[k] = v

        def clone(self):
            return self

    cfg.CfgNode = CfgNode
    yacs.config = cfg
    sys.modules["yacs"] = yacs
    sys.modules["yacs.config"] = cfg
    sys.path.insert(0, REF)
    os.chdir(os.path.join(REF, "QFA"))
    import QFA  # noqa
    from QFA import model, optimizer, utils
    os.chdir(REPO)
    return model, optimizer, utils


def main():
    import torch
    torch.manual_seed(0)
    model, optimizer, utils = import_reference()
    from qfa_amd import synthetic

    dev = torch.device("cpu")
    wav, nb, nr = synthetic.wavelength_grid()
    assert (len(wav), nb) == (1913, 720)

    def T(x, dt=torch.float32):
        return torch.tensor(np.asarray(x), dtype=dt)

    def npd(d):
        return {k: v.detach().numpy() for k, v in d.items()}

    # ---- copied data fixtures (inputs) -----------------------------------------------------
    spec = np.load(os.path.join(REF, "data", "spec-4321-55504-0114.npz"))
    np.savez_compressed(os.path.join(HERE, "sdss_spectrum.npz"),
                        flux=spec["flux"], error=spec["error"], z=spec["z"],
                        ll=spec["ll"], h=spec["h"], our=spec["our"],
                        our_uncertainty=spec["our_uncertainty"],
                        ll_red=spec["ll_red"], h_red=spec["h_red"], our_red=spec["our_red"])
    shutil.copyfile(os.path.join(REF, "data", "model_parameters.npz"),
                    os.path.join(HERE, "model_parameters.npz"))
    os.chmod(os.path.join(HERE, "model_parameters.npz"), 0o644)

    # ---- G1 / G2 ----------------------------------------------------------------------------
    m = model.QFA(nb, nr, 8, dev)
    m.load_from_npz(os.path.join(REF, "data", "model_parameters.npz"))
    flux, error, z = spec["flux"], spec["error"], float(spec["z"])
    mask = (flux != -999.) & (error != -999.)
    zabs = wav[:nb] * (1 + z) / 1215.67 - 1
    out = {}
    for tag, mk in (("full", mask), ("red", mask & (np.arange(len(wav)) >= nb))):
        ll, hm, hc, cont, unc = m.prediction_for_single_spectra(T(flux), T(error), T(zabs), T(mk, torch.bool))
        out.update({f"ll_{tag}": ll.numpy(), f"hmean_{tag}": hm.numpy(), f"hcov_{tag}": hc.numpy(),
                    f"cont_{tag}": cont.numpy(), f"unc_{tag}": unc.numpy(), f"mask_{tag}": mk})
    np.savez_compressed(os.path.join(HERE, "g1_g2_predict.npz"), zabs=zabs.astype(np.float32), **out)

    shipped = {k: getattr(m, k).numpy().copy() for k in ("F", "Psi", "omega", "tau0", "c0", "beta")}
    mu = m.mu.numpy().copy()

    # ---- G3 ---------------------------------------------------------------------------------
    b = synthetic.make_batch_numpy(shipped, mu, wav, nb, 4, seed=20220703)
    nlls, gs = [], []
    for s in range(4):
        nll, g = m.loglikelihood_and_gradient_for_single_spectra(
            T(b["delta"][s]), T(b["error"][s]), T(b["zabs"][s]), T(b["mask"][s], torch.bool))
        nlls.append(nll.numpy().squeeze())
        gs.append(npd(g))
    np.savez_compressed(os.path.join(HERE, "g3_single.npz"), seed=20220703, nll=np.array(nlls),
                        **{f"g_{k}": np.stack([g[k] for g in gs]) for k in gs[0]})

    # ---- G4 ---------------------------------------------------------------------------------
    b = synthetic.make_batch_numpy(shipped, mu, wav, nb, 8, seed=20220704, red_only=(3,), dead_range=(900, 910))
    loss, g = m.forward(T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"], torch.bool))
    np.savez_compressed(os.path.join(HERE, "g4_forward.npz"), seed=20220704, loss=loss.numpy(),
                        **{f"g_{k}": v for k, v in npd(g).items()})

    # ---- G5 ---------------------------------------------------------------------------------
    b = synthetic.make_batch_numpy(shipped, mu, wav, nb, 128, seed=20220701)
    m5 = model.QFA(nb, nr, 8, dev)
    m5.load_from_npz(os.path.join(REF, "data", "model_parameters.npz"))
    opt = optimizer.Adam(params=m5.parameters, device=dev, scheduler=optimizer.step_scheduler(0.9, 10),
                         learning_rate=1e-3, weight_decay=1e-1)
    loss, g = m5.forward(T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"], torch.bool))
    m5.parameters = opt.update(m5.parameters, g)
    np.savez_compressed(os.path.join(HERE, "g5_step.npz"), seed=20220701, loss=loss.numpy(),
                        **{f"g_{k}": v for k, v in npd(g).items()},
                        **{f"p_{k}": v for k, v in npd(m5.parameters).items()})

    # ---- G6 ---------------------------------------------------------------------------------
    m6 = model.QFA(nb, nr, 8, dev)
    m6.load_from_npz(os.path.join(REF, "data", "model_parameters.npz"))
    m6.smooth()
    sm = npd(m6.parameters)
    m6.Psi = m6.Psi * 30.0 - 1.0
    m6.omega = m6.omega * 10.0 - 0.5
    m6.tau0 = m6.tau0 * 0 + 1.7
    m6.beta = m6.beta * 0 + 0.01
    m6.c0 = m6.c0 * 0 - 9.0
    pre = npd(m6.parameters)
    m6.clip()
    np.savez_compressed(os.path.join(HERE, "g6_smooth_clip.npz"),
                        **{f"smooth_{k}": v for k, v in sm.items()},
                        **{f"preclip_{k}": v for k, v in pre.items()},
                        **{f"clip_{k}": v for k, v in npd(m6.parameters).items()})

    # ---- G7 ---------------------------------------------------------------------------------
    rng = np.random.default_rng(7)
    p0 = {"F": rng.standard_normal((12, 3)).astype(np.float32), "Psi": rng.random(12).astype(np.float32),
          "omega": rng.random(5).astype(np.float32), "tau0": np.float32(0.02), "c0": np.float32(0.3),
          "beta": np.float32(2.0)}
    grads = [{k: (rng.standard_normal(np.shape(v)) * 0.1).astype(np.float32) for k, v in p0.items()}
             for _ in range(6)]
    params = {k: T(v) for k, v in p0.items()}
    opt = optimizer.Adam(params=params, device=dev, scheduler=optimizer.step_scheduler(0.9, 2),
                         learning_rate=1e-2, weight_decay=1e-3)
    trace = {}
    it = 0
    for epoch in range(3):
        for _ in range(2):
            params = opt.update(params, {k: T(v) for k, v in grads[it].items()})
            for k, v in params.items():
                trace[f"p{it}_{k}"] = v.numpy().copy()
            it += 1
        opt.step()
    np.savez_compressed(os.path.join(HERE, "g7_adam.npz"),
                        **{f"init_{k}": v for k, v in p0.items()},
                        **{f"grad{i}_{k}": v for i, g in enumerate(grads) for k, v in g.items()}, **trace)

    # ---- G8 ---------------------------------------------------------------------------------
    M = rng.standard_normal((64, 4)).astype(np.float32)
    D = (0.5 + rng.random(64)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g8_woodbury.npz"), M=M, D=D,
                        inv=utils.MatrixInverse(T(M), T(D), dev).numpy(),
                        logdet=utils.MatrixLogDet(T(M), T(D), dev).numpy())

    # ---- G9 ---------------------------------------------------------------------------------
    zg = np.linspace(1.5, 4.0, 41).astype(np.float32)
    d9 = {"z": zg}
    for which in ("becker", "fg", "kamble", "mock"):
        for series in (1, 2, 5):
            d9[f"tau_{which}_{series}"] = utils.tau(T(zg), which=which, series=series).numpy()
    d9["tauHI"] = utils.tauHI(T(zg), T(0.0123), T(3.1)).numpy()
    d9["omega_func"] = utils.omega_func(T(zg), T(0.0123), T(3.1), T(0.27)).numpy()
    np.savez_compressed(os.path.join(HERE, "g9_tau.npz"), **d9)

    # ---- G10 --------------------------------------------------------------------------------
    # random_init_func-like parameters (QFA/model.py:67-72) at N_pix=4000, N_h=16: the reference's
    # float32 det(I + M^T D^-1 M) overflows (QFA/utils.py:54) and its loss is +inf (quirk Q7);
    # the gradients do not involve the determinant and stay finite.
    wav16, nb16, nr16 = synthetic.wavelength_grid(4000)
    r16 = np.random.default_rng(16)
    p16 = {"F": (r16.random((4000, 16)) - 0.5).astype(np.float32), "Psi": np.ones(4000, np.float32),
           "omega": np.ones(nb16, np.float32), "tau0": np.float32(0.02), "c0": np.float32(0.3),
           "beta": np.float32(2.0)}
    _, mu16 = synthetic.mock_parameters(4000, nb16, 16, seed=16)
    b = synthetic.make_batch_numpy(p16, mu16, wav16, nb16, 2, seed=20220710)
    m16 = model.QFA(nb16, nr16, 16, dev, model_params=p16)
    loss, g = m16.forward(T(b["delta"]), T(b["error"]), T(b["zabs"]), T(b["mask"], torch.bool))
    np.savez_compressed(os.path.join(HERE, "g10_k16.npz"), seed=20220710, n_pix=4000, loss=loss.numpy(),
                        **{f"g_{k}": v for k, v in npd(g).items()})
    print("reference k=16 float32 loss:", loss)

    # ---- G11: host preprocessing of the dataloader (next row N1) -------------------------------
    # tau_total / smooth come from the imported reference (QFA/utils.py:174-219); the three
    # expressions around them are the call sites QFA/dataloader.py:102, 110-112, 135-138.
    r11 = np.random.default_rng(11)
    zq = r11.uniform(2.0, 3.5, size=6)
    wav_lo = 10 ** np.arange(np.log10(1000.0), np.log10(1600.0), 1e-4)     # reaches below Ly-beta: 2 series
    out11 = {"zqso": zq}
    for tag, wv in (("c1", wav), ("lyb", wav_lo)):
        nbb = int(np.sum(wv < 1215.67))
        fl = 1.0 + 0.3 * r11.standard_normal((6, len(wv)))
        mk = r11.random((6, len(wv))) > 0.05
        fl = np.where(mk, fl, -999.0)
        for which in ("becker", "kamble"):
            tt = utils.tau_total(wv, zq, which=which)
            out11[f"tau_total_{tag}_{which}"] = tt
        tt = utils.tau_total(wv, zq, which="becker")
        zabs11 = (zq + 1).reshape(-1, 1) * wv[:nbb] / 1215.67 - 1
        s_up = np.hstack((np.exp(1 * tt), np.ones((6, len(wv) - nbb), dtype=float)))
        mu_raw = np.sum(fl * s_up * mk, axis=0) / np.sum(fl != -999., axis=0)
        mu_s = utils.smooth(mu_raw, window_len=16)
        s_dn = np.hstack((np.exp(-1 * tt), np.ones((6, len(wv) - nbb), dtype=float)))
        delta11 = fl - mu_s * s_dn
        out11.update({f"wav_{tag}": wv, f"flux_{tag}": fl, f"zabs_{tag}": zabs11, f"mu_raw_{tag}": mu_raw,
                      f"mu_{tag}": mu_s, f"delta_{tag}": delta11})
    np.savez_compressed(os.path.join(HERE, "g11_dataprep.npz"), **out11)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
