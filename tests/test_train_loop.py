"""QFA.train control flow on the GPU against an oracle-driven replica of the reference loop
(reference QFA/model.py:183-231): Niter = N // B with a trailing partial batch, Adam.step() once
per epoch, smooth / save cadence, early stop on negative epoch loss, .npz round trip with the
c0 <- beta load quirk."""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
KEYS = ("F", "Psi", "omega", "tau0", "c0", "beta")


class FakeLoader:
    """Dataloader contract consumed by QFA.train (reference QFA/dataloader.py:114-138,154-167,189-191),
    deterministic (rewind does not shuffle)."""

    def __init__(self, b, mu, batch_size, device):
        import torch
        self.t = {k: torch.tensor(b[k], device=device) for k in ("delta", "error", "zabs", "mask")}
        self.mu = mu
        self.data_size = b["delta"].shape[0]
        self.batch_size = batch_size
        self.cur = 0

    def rewind(self):
        self.cur = 0

    def have_next_batch(self):
        return self.cur < self.data_size

    def next_batch(self):
        s, e = self.cur, min(self.cur + self.batch_size, self.data_size)
        self.cur = e
        return tuple(self.t[k][s:e] for k in ("delta", "error", "zabs", "mask"))


def _oracle_train(p, b, batch_size, n_epochs, lr, alpha, step, wd, smooth_interval):
    from oracle import qfa_oracle as O
    params = {k: np.asarray(v, dtype=np.float64) for k, v in p.items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v = {k: np.zeros_like(vv) for k, vv in params.items()}
    N = b["delta"].shape[0]
    niter = N // batch_size
    losses = []
    i = 0
    for epoch in range(n_epochs):
        tot = 0.0
        for s in range(0, N, batch_size):
            e = min(s + batch_size, N)
            loss, g = O.forward(params, b["delta"][s:e], b["error"][s:e], b["zabs"][s:e], b["mask"][s:e])
            tot += loss / niter
            params, m, v = O.adam_update(m, v, i, params, g, O.step_lr(i, lr, alpha, step), weight_decay=wd)
            params = O.clip_params(params)
        i += 1
        losses.append(tot)
        if tot < 0:
            params = O.smooth_params(params)
            break
        if (epoch + 1) % smooth_interval == 0:
            params = O.smooth_params(params)
    return params, losses


def test_train_two_epochs_matches_oracle_loop(tmp_path, capsys):
    import torch
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(320)
    p, mu = synthetic.mock_parameters(320, nb, 4, seed=21)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 22, seed=211)           # 22 = 2 x 8 + 6: partial last batch
    model = QFA(nb, nr, 4, dev, model_params=p)
    opt = Adam(model.parameters, dev, scheduler=step_scheduler(0.9, 1), learning_rate=1e-3, weight_decay=1e-1)
    model.train(opt, FakeLoader(b, mu, 8, dev), 2, str(tmp_path), save_interval=1, smooth_interval=2, quiet=False)
    out = capsys.readouterr().out
    ref, losses = _oracle_train(p, b, 8, 2, 1e-3, 0.9, 1, 1e-1, 2)
    printed = [float(line.split("loss:")[1].split(";")[0]) for line in out.splitlines() if "loss:" in line]
    assert len(printed) == 2
    for a, r in zip(printed, losses):
        assert abs(a - r) <= 0.006 + 1e-5 * abs(r)                          # printed with 2 decimals
    assert opt.i == 2
    for k in KEYS:
        assert rel_l2(model.parameters[k].cpu().numpy(), ref[k]) < 2e-5, k
    ck = os.path.join(str(tmp_path), "checkpoints")
    assert sorted(os.listdir(ck)) == ["model_parameters_epoch_01.npz", "model_parameters_epoch_02.npz"]
    f = np.load(os.path.join(ck, "model_parameters_epoch_02.npz"))
    assert sorted(f.files) == sorted(["mu", "F", "Psi", "omega", "tau0", "c0", "beta"])
    # load round trip: quirk on (reference behaviour) and off
    m2 = QFA(nb, nr, 4, dev)
    m2.load_from_npz(os.path.join(ck, "model_parameters_epoch_02.npz"))
    assert torch.equal(m2.F, model.F) and torch.equal(m2.c0, model.beta)      # c0 <- beta (model.py:295)
    m2.load_from_npz(os.path.join(ck, "model_parameters_epoch_02.npz"), reference_c0_quirk=False)
    assert torch.equal(m2.c0, model.c0)
    assert torch.equal(m2.mu, torch.tensor(mu, device=dev))


def test_train_early_stop_and_zero_niter(tmp_path):
    import torch
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(200)
    p, mu = synthetic.mock_parameters(200, nb, 4, seed=3)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 6, seed=33, masks=False)  # (a pixel masked in a whole batch
    b["error"] = (b["error"] * 0 + 0.01).astype(np.float32)                 #  would poison the parameters with NaN,
    b["delta"] = (b["delta"] * 1e-3).astype(np.float32)                     #  as in the reference)
    p2 = dict(p)
    p2["Psi"] = np.full_like(p["Psi"], 1e-3)
    p2["omega"] = np.full_like(p["omega"], 1e-3)
    p2["F"] = (p["F"] * 1e-3).astype(np.float32)
    model = QFA(nb, nr, 4, dev, model_params=p2)
    opt = Adam(model.parameters, dev, scheduler=None, learning_rate=1e-6, weight_decay=0.0)
    model.train(opt, FakeLoader(b, mu, 3, dev), 5, str(tmp_path), save_interval=5, smooth_interval=5, quiet=True)
    # epoch-mean NLL is negative at once: exactly one epoch ran, smoothed and saved (model.py:224-227)
    assert opt.i == 1
    assert os.listdir(os.path.join(str(tmp_path), "checkpoints")) == ["model_parameters_epoch_01.npz"]
    with pytest.raises(ZeroDivisionError):                                   # Niter = 6 // 8 = 0 (model.py:205,213)
        model.train(opt, FakeLoader(b, mu, 8, dev), 1, str(tmp_path), quiet=True)


def test_adam_state_dict_roundtrip():
    import torch
    from qfa_amd import Adam
    dev = torch.device("cuda:0")
    params = {"F": torch.randn(5, 2, device=dev), "Psi": torch.rand(5, device=dev)}
    g = {k: torch.randn_like(v) for k, v in params.items()}
    a = Adam(params, dev, scheduler=None)
    a.update(params, g); a.step()
    b = Adam(params, dev, scheduler=None)
    b.load_state_dict(a.state_dict())
    out_a, out_b = a.update(params, g), b.update(params, g)
    for k in params:
        assert torch.equal(out_a[k], out_b[k])


def test_predict_writer_and_checkpoint_resume(tmp_path):
    """Row N2: per-spectrum .npz of main.py:94-98 (batched) and a checkpoint that carries the Adam state."""
    import torch
    from oracle import qfa_oracle as O
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    from qfa_amd.dataloader import DeviceDataloader
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(256)
    p, mu0 = synthetic.mock_parameters(256, nb, 4, seed=8)
    b = synthetic.make_batch_numpy(p, mu0, wav, nb, 9, seed=81)
    paths = [f"/data/spec-{i:04d}.npz" for i in range(9)]
    dl = DeviceDataloader(b["flux"], b["error"], b["zqso"], wav, batch_size=4, device=dev, shuffle=False, paths=paths)
    model = QFA(nb, nr, 4, dev, model_params=p)
    model.mu = torch.tensor(dl.mu, dtype=torch.float32, device=dev)
    out = tmp_path / "predict"
    names = model.predict_to_npz(dl, str(out), batch_size=4)
    assert names == [f"spec-{i:04d}.npz" for i in range(9)]
    mu32 = model.mu.cpu().numpy()
    for i in (0, 5, 8):
        f = np.load(out / names[i])
        assert sorted(f.files) == ["cont", "hcov", "hmean", "ll", "uncertainty"]
        assert f["ll"].shape == (1, 1) and f["hmean"].shape == (4, 1) and f["hcov"].shape == (4, 4)
        zabs = O.zabs_from_zqso(wav, b["zqso"][i:i + 1], nb)[0].astype(np.float32)
        o = O.predict_single(p, mu32, b["flux"][i], b["error"][i], zabs, b["mask"][i])
        assert abs(f["ll"].item() - o[0]) / abs(o[0]) < 1e-5
        assert rel_l2(f["cont"], o[3]) < 1e-4 and rel_l2(f["uncertainty"], o[4]) < 1e-4
    # checkpoint with optimiser state: a resumed run reproduces the uninterrupted one (up to the
    # re-association of the float atomics between two launches)
    T = lambda k: torch.tensor(b[k], device=dev)
    args = (T("delta"), T("error"), T("zabs"), T("mask"))
    m1 = QFA(nb, nr, 4, dev, model_params=p)
    o1 = Adam(m1.parameters, dev, scheduler=step_scheduler(0.9, 1), learning_rate=1e-3, weight_decay=1e-1)
    m1.step(o1, *args); o1.step()
    m1.save_checkpoint(str(tmp_path / "ck.npz"), o1)
    m1.step(o1, *args)
    m2 = QFA(nb, nr, 4, dev)
    o2 = Adam(m2.parameters, dev, scheduler=step_scheduler(0.9, 1), learning_rate=1e-3, weight_decay=1e-1)
    m2.load_checkpoint(str(tmp_path / "ck.npz"), o2)
    assert o2.i == 1
    m2.step(o2, *args)
    for k in KEYS:
        assert rel_l2(m2.parameters[k].cpu().numpy(), m1.parameters[k].cpu().numpy()) < 1e-6, k


@pytest.mark.parametrize("loader_kind", ["device", "foreign"])
def test_train_with_step_graph_matches_eager(tmp_path, loader_kind):
    """hipGraph replay of the step (StepGraph) against the eager loop: same parameters after three epochs with
    a trailing partial batch, smoothing in between and a learning-rate change per epoch (re-capture)."""
    import torch
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    from qfa_amd.dataloader import DeviceDataloader
    from qfa_amd.model import StepGraph
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(320)
    p, mu0 = synthetic.mock_parameters(320, nb, 4, seed=5)
    b = synthetic.make_batch_numpy(p, mu0, wav, nb, 27, seed=51, masks=False)

    def run(use_graph):
        if loader_kind == "device":
            dl = DeviceDataloader(b["flux"], b["error"], b["zqso"], wav, batch_size=6, device=dev, shuffle=False)
        else:
            dl = FakeLoader(b, mu0, 6, dev)
        model = QFA(nb, nr, 4, dev, model_params=p)
        opt = Adam(model.parameters, dev, scheduler=step_scheduler(0.5, 1), learning_rate=1e-3, weight_decay=1e-1)
        model.train(opt, dl, 3, str(tmp_path / ("g" if use_graph else "e")), quiet=True, smooth_interval=2,
                    use_graph=use_graph)
        return {k: model.parameters[k].cpu().numpy() for k in KEYS}

    eager, graph = run(False), run(True)
    for k in KEYS:
        assert rel_l2(graph[k], eager[k]) < 1e-6, k
    # the graph really replays: 4 full batches per epoch, the first one of an epoch eager, the second captured
    model = QFA(nb, nr, 4, dev, model_params=p)
    opt = Adam(model.parameters, dev, learning_rate=1e-3)
    sg = StepGraph(model, opt, 6)
    dl = FakeLoader(b, mu0, 6, dev)
    n = 0
    while dl.have_next_batch() and sg.fits(dl) and n < 4:
        loss = sg.run_next(dl)
        n += 1
    torch.cuda.synchronize()
    assert sg.replays == 3 and torch.isfinite(loss).all()


@pytest.mark.gpu
def test_model_without_blue_pixels_trains(tmp_path):
    """N_b = 0 (omega is an empty tensor with a NULL data pointer): forward supported it, step / clip / smooth /
    train must too."""
    import torch
    from oracle import qfa_oracle as O
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    dev = torch.device("cuda:0")
    npix, nh, B = 96, 3, 9
    rng = np.random.default_rng(3)
    p = {"F": (0.3 * rng.standard_normal((npix, nh))).astype(np.float32), "Psi": np.full(npix, 0.1, np.float32),
         "omega": np.zeros(0, np.float32), "tau0": np.float32(0.1), "c0": np.float32(0.2), "beta": np.float32(1.5)}
    d = rng.standard_normal((B, npix)).astype(np.float32)
    e = (0.1 + 0.1 * rng.random((B, npix))).astype(np.float32)
    mk = rng.random((B, npix)) > 0.05
    mk[0] = True                                      # every pixel is observed in every batch below (batches start at rows 0, 4, 8:
    mk[4] = True                                      # a pixel masked in a whole batch gets a 0/0 = NaN gradient, quirk Q3)
    mk[8] = True
    z = np.zeros((B, 0), np.float32)
    T = lambda x: torch.tensor(x, device=dev)
    m = QFA(0, npix, nh, dev, model_params=p)
    opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)
    loss, g = m.forward(T(d), T(e), T(z), T(mk))
    oloss, og = O.forward(p, d, e, z, mk)
    assert abs(loss.item() - oloss) / abs(oloss) < 1e-5
    assert g["omega"].numel() == 0
    m.step(opt, T(d), T(e), T(z), T(mk))
    m.clip()
    m.smooth()
    assert m.omega.numel() == 0 and torch.isfinite(m.F).all() and torch.isfinite(m.Psi).all()

    class L:                                          # a foreign loader with the reference's contract
        mu, data_size, batch_size = np.ones(npix, np.float32), B, 4
        def rewind(self): self.cur = 0
        def have_next_batch(self): return self.cur < B
        def next_batch(self):
            a, self.cur = self.cur, min(self.cur + 4, B)
            return T(d[a:self.cur]), T(e[a:self.cur]), T(z[a:self.cur]), T(mk[a:self.cur])
    m.train(opt, L(), 2, str(tmp_path), quiet=True)
    assert torch.isfinite(m.F).all()


@pytest.mark.gpu
def test_step_graph_key_follows_adam_buffers():
    """Adam.reset / load_state_dict allocate new moment tensors: the captured graph must not replay into the old ones"""
    import torch
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(128)
    p, mu = synthetic.mock_parameters(128, nb, 4, seed=2)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 8, seed=22)
    m = QFA(nb, nr, 4, dev, model_params=p)
    opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 10), learning_rate=1e-3, weight_decay=1e-1)
    sg = m.step_graph(opt, 8)
    for dst, key in zip(sg.buf, ("delta", "error", "zabs", "mask")):
        dst.copy_(torch.tensor(b[key], device=dev))
    for _ in range(3):
        sg.run()
    assert sg.replays >= 1
    k0 = sg._key()
    opt.reset(m.parameters)
    assert sg._key() != k0
    sg.run()                                          # eager step with the new buffers, then re-capture
    sg.run()
    torch.cuda.synchronize()
    assert all(float(opt.m[k].abs().sum()) > 0 for k in ("F", "Psi"))      # the LIVE moments were updated


@pytest.mark.parametrize("npix,nh,flags", [(640, 16, 0), (704, 24, 0), (640, 12, 0),
                                           (640, 16, 0x40), (640, 8, 0x40)])      # 0x40 = F_PASS2_PIXRES: k_solve's operand images + k_grads_t
def test_training_trajectory_on_the_xdl_path_matches_oracle_loop(tmp_path, npix, nh, flags):
    """north_star: "learned F / Psi / mu within a stated fp32 tolerance" -- on the kernels the headline runs on
    (N_h = 9..16: k_moments_x + k_grads_x; 17..32: k_moments_x<32> + k_s12_x + k_grads_s3), not only the N_h = 4 case above.
    3 epochs x 4 batches (one partial) against the oracle-driven replica of QFA/model.py:204-215; tolerance 2e-5 on
    every learned tensor."""
    import torch
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=60 + nh)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, 120, seed=61 + nh)       # 120 = 3 x 32 + 24
    model = QFA(nb, nr, nh, dev, model_params=p)
    model.flags = flags
    opt = Adam(model.parameters, dev, scheduler=step_scheduler(0.9, 1), learning_rate=1e-3, weight_decay=1e-1)
    model.train(opt, FakeLoader(b, mu, 32, dev), 3, str(tmp_path), save_interval=10, smooth_interval=2, quiet=True,
                use_graph=bool(flags))       # (the pixel-resident cases also replay the step as a hipGraph)
    ref, losses = _oracle_train(p, b, 32, 3, 1e-3, 0.9, 1, 1e-1, 2)
    assert opt.i == 3
    for k in KEYS:
        assert rel_l2(model.parameters[k].cpu().numpy(), ref[k]) < 2e-5, (k, rel_l2(model.parameters[k].cpu().numpy(), ref[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("npix,nh,B", [(320, 4, 27), (640, 12, 70), (97, 20, 9)])
def test_fused_finalize_adam_is_bit_identical_to_the_two_calls(npix, nh, B):
    """qfa_finalize_adam_clip_f32 (QFA.step: sum / count and Adam + clip in one launch) against qfa_finalize_grads_f32 followed by
    qfa_adam_clip_multi_f32 (QFA.forward + Adam.update, reference QFA/model.py:212-214): the same bits in the new parameters,
    both moments and the loss -- a dead pixel range (count 0: NaN gradients, quirk Q3) included"""
    import torch
    from qfa_amd import QFA, Adam, step_scheduler, synthetic
    dev = torch.device("cuda:0")
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu0 = synthetic.mock_parameters(npix, nb, nh, seed=5)
    b = synthetic.make_batch_numpy(p, mu0, wav, nb, B, seed=51, dead_range=(40, 44))
    T = lambda k: torch.tensor(b[k], device=dev)
    res = []
    for fused in (False, True):
        m = QFA(nb, nr, nh, dev, model_params=p)
        opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.5, 1), learning_rate=1e-3, weight_decay=1e-1)
        first = None
        for it in range(3):
            if fused:
                loss = m.step(opt, T("delta"), T("error"), T("zabs"), T("mask"))
            else:
                loss, g = m.forward(T("delta"), T("error"), T("zabs"), T("mask"))
                new = opt.update(m.parameters, g, clip=m._clip_table())
                for k in KEYS:
                    setattr(m, k, new[k])
            first = loss.clone() if first is None else first        # (from the second step on F holds NaN rows: NaN loss)
            opt.step()
        res.append(([getattr(m, k).clone() for k in KEYS], [opt.m[k].clone() for k in KEYS], [opt.v[k].clone() for k in KEYS], first))
    for a, c in zip(res[0][:3], res[1][:3]):
        for x, y in zip(a, c):
            assert torch.equal(torch.nan_to_num(x, nan=123.0), torch.nan_to_num(y, nan=123.0))
            assert torch.equal(torch.isnan(x), torch.isnan(y))
    assert torch.equal(res[0][3], res[1][3]) and torch.isfinite(res[0][3]).all()
    assert torch.isnan(res[0][0][0][40:44]).all()                          # the dead pixels' F rows went NaN in both
