"""The resident, indexed input form (ABI v3: qfa_batch_t.rows / row_stride; qfa_amd/resident.py) on the GPU.

What the reference does per batch on the host (QFA/dataloader.py:124-138: materialise delta / error / zabs / mask of the
batch; :154-167: shuffle by permuting the data set) is here an index array into arrays that stay where they are.  The
tests require the indexed form to be BIT-IDENTICAL to the same rows gathered into contiguous tensors -- every pass-1 /
pass-2 form, every N_h range, zabs and factored z, padded and unpadded rows, rows beyond a 4-GiB offset -- and the
loader / train / step-graph paths built on it to reproduce the materialised ones."""
import numpy as np
import pytest

from conftest import rel_l2
from qfa_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def resident_set(dev, npix, nh, N, seed, stride=None, with_zabs=False):
    """N mock spectra as resident arrays with the given row stride (default: padded to 32 pixels)"""
    import torch
    from qfa_amd import QFA, synthetic
    from qfa_amd.resident import ResidentBatch
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=seed)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, N, seed=seed + 1)
    stride = (npix + 31) // 32 * 32 if stride is None else stride
    def pad(a, dt, fill):
        t = torch.full((N, stride), fill, dtype=dt, device=dev)            # (pad pixels hold garbage on purpose: never read)
        t[:, :npix] = torch.as_tensor(a, device=dev).to(dt)
        return t
    f32 = torch.float32
    flux, delta, error = pad(b["flux"], f32, 7.5e8), pad(b["delta"], f32, -3.25e9), pad(b["error"], f32, float("nan"))
    mask = pad(b["mask"], torch.bool, True)
    zq1 = torch.as_tensor((1.0 + b["zqso"].astype(np.float64)).astype(np.float32), device=dev)
    ratio = torch.as_tensor((wav[:nb] / synthetic.LYA).astype(np.float32), device=dev)
    zabs = torch.as_tensor(b["zabs"], device=dev).to(f32).contiguous() if with_zabs else None
    m = QFA(nb, nr, nh, dev, model_params=p)
    m.mu = torch.as_tensor(mu, device=dev).to(f32)
    rb = ResidentBatch(flux, delta, error, mask, None if with_zabs else zq1, None if with_zabs else ratio, None, npix, nb,
                       zabs=zabs)
    return m, rb, b, p, mu


def perm_rows(dev, N, B, seed):
    import torch
    rng = np.random.default_rng(seed)
    return torch.as_tensor(rng.permutation(N)[:B].astype(np.int32), device=dev)


CASES = [
    # npix, nh, N, B, flags, zabs form
    (200, 16, 150, 70, 0, False), (200, 16, 150, 70, 0, True),                      # k_moments_x<16> + k_grads_x<16>
    (97, 9, 80, 33, _lib.F_PASS2_PIXRES, False), (1913, 12, 900, 700, _lib.F_PASS2_PIXRES, True),     # k_grads_t<16>: ragged / straddling tiles
    (1913, 8, 400, 130, 0, False), (1913, 8, 400, 130, _lib.F_PASS2_PIXRES, False), (1000, 8, 300, 130, _lib.F_PASS2_PIXRES, True),
    (97, 5, 60, 33, _lib.F_PASS2_PIXRES, False), (450, 1, 100, 65, 0, True),
    (640, 16, 100, 48, _lib.F_PASS2_F32, False), (640, 16, 100, 48, _lib.F_PASS2_F32, True), (200, 8, 100, 70, _lib.F_PASS2_F32, False),
    (450, 32, 120, 70, 0, False), (1000, 20, 200, 130, 0, True), (31, 17, 9, 5, 0, False),             # k_moments_x<32>, k_s12_x
    (4000, 16, 2000, 1100, _lib.F_PASS2_PIXRES, False), (64, 16, 3, 1, 0, False),
]


@pytest.mark.parametrize("npix,nh,N,B,flags,zform", CASES)
def test_indexed_batch_is_bit_identical_to_the_gathered_batch(dev, npix, nh, N, B, flags, zform):
    """forward (deterministic accumulation: fixed summation order) and predict on rows picked by a random permutation out of a
    resident set with padded rows, against the same rows gathered into contiguous tensors"""
    import torch
    m, rb0, b, p, mu = resident_set(dev, npix, nh, N, seed=3 * npix + nh, with_zabs=zform)
    rb = rb0.with_rows(perm_rows(dev, N, B, seed=npix + B))
    (d, e, z, mk), zfac = rb.materialize()
    m.flags, m.deterministic = flags, True
    nll_i = torch.empty(B, dtype=torch.float32, device=dev)
    nll_g = torch.empty(B, dtype=torch.float32, device=dev)
    acc_i = m.accumulate(batch=rb, nll=nll_i).clone()
    acc_g = m.accumulate(d, e, z, mk, zfac=zfac, nll=nll_g).clone()
    torch.cuda.synchronize()
    assert torch.equal(nll_i, nll_g)
    assert torch.equal(acc_i, acc_g)
    assert torch.isfinite(acc_i).all()                                      # (the garbage in the pad pixels was never read)
    # the gathered batch against the oracle: the indexed form is not compared with itself only
    from oracle import qfa_oracle as O
    rows = rb.rows.cpu().numpy()
    oloss, og = O.forward(p, b["delta"][rows], b["error"][rows], b["zabs"][rows], b["mask"][rows])
    loss, g = m._finalize(acc_i, True)
    assert abs(loss.item() - oloss) / abs(oloss) < 5e-6
    ref = np.asarray(og["Psi"])
    ok = ~np.isnan(ref)
    assert rel_l2(g["Psi"].cpu().numpy()[ok], ref[ok]) < 2e-5
    # predict: raw flux rows
    (fx, e2, z2, mk2), zfac2 = rb.materialize(raw_flux=True)
    m.flags = 0
    out_i = m.predict(batch=rb)
    out_g = m.predict(fx, e2, z2, mk2, zfac=zfac2)
    for a, c in zip(out_i, out_g):
        assert torch.equal(a, c)


POISON_CASES = [
    # npix, nh, N, flags, zabs form
    (200, 16, 70, 0, False), (200, 16, 70, 0, True), (1913, 12, 300, _lib.F_PASS2_PIXRES, True), (1913, 12, 300, _lib.F_PASS2_PIXRES, False),
    (1913, 8, 200, 0, False), (1000, 8, 200, _lib.F_PASS2_PIXRES, False), (1000, 8, 200, _lib.F_PASS2_PIXRES, True),
    (640, 16, 48, _lib.F_PASS2_F32, False), (450, 24, 70, 0, False), (450, 24, 70, 0, True),
]


@pytest.mark.parametrize("npix,nh,N,flags,zform", POISON_CASES)
def test_masked_pixels_may_hold_anything(dev, npix, nh, N, flags, zform):
    """The reference marks bad pixels with -999 (QFA/dataloader.py:24,28) and never reads them; here a masked pixel may hold
    NaN, +-inf or the largest float32 in delta, flux AND error: every sum, every per-spectrum NLL and every predicted value is
    bit for bit what the sentinel batch gives (the kernels select or clamp, they do not multiply a NaN by zero)."""
    import torch
    m, rb0, b, p, mu = resident_set(dev, npix, nh, N, seed=5 * npix + nh, with_zabs=zform)
    rb = rb0.with_rows(torch.arange(N, dtype=torch.int32, device=dev))
    g = torch.Generator(device="cpu").manual_seed(npix + nh)
    rb.mask[:, :npix] &= (torch.rand((N, npix), generator=g) > 0.05).to(dev)      # (5 % more masked pixels, scattered)
    rb.mask[N // 2, :npix] = False                                                # and one spectrum without a valid pixel
    m.flags, m.deterministic = flags, True
    nll0 = torch.empty(N, dtype=torch.float32, device=dev)
    acc0 = m.accumulate(batch=rb, nll=nll0).clone()
    out0 = [x.clone() for x in m.predict(batch=rb)]
    bad = ~rb.mask[:, :npix]
    assert bad.any()
    kind = torch.randint(0, 5, bad.shape, generator=g).to(dev)
    poison = torch.tensor([float("nan"), float("inf"), -float("inf"), 3.4028234e38, -3.4028234e38], device=dev)[kind]
    for arr in (rb.delta, rb.flux, rb.error):
        v = arr[:, :npix]
        v[bad] = poison[bad]
    nll1 = torch.empty(N, dtype=torch.float32, device=dev)
    acc1 = m.accumulate(batch=rb, nll=nll1).clone()
    out1 = m.predict(batch=rb)
    torch.cuda.synchronize()
    assert torch.equal(nll0, nll1) and torch.isfinite(nll1).all()
    assert torch.equal(acc0, acc1)
    for a, c in zip(out0, out1):
        assert torch.equal(a, c)
    # and through the tensor form (gathered copies of the poisoned rows)
    (d, e, z, mk), zfac = rb.materialize()
    acc2 = m.accumulate(d, e, z, mk, zfac=zfac).clone()
    assert torch.equal(acc0, acc2)


@pytest.mark.parametrize("npix,nh,masked,zform", [(1913, 8, True, False), (1913, 8, False, False), (1913, 12, True, True), (9243 // 8, 8, True, False),
                                                  (77, 20, True, False)])
def test_zero_error_in_the_last_pixel_of_a_ragged_row(dev, npix, nh, masked, zform):
    """the pad pixels of the ragged last 32-pixel tile (N_pix = 1913, 9243: the reference's shapes) clone delta / sigma of
    pixel N_pix - 1 and have Psi = 0 in pass 1's image; with error == 0 there (the pixel masked or not) D was 0, 1/D inf and the
    mask FACTOR 0 made a NaN of it -- in every moment and the NLL of the spectrum.  The reference handles such a pixel (it is
    simply masked, or its D is Psi A^2 > 0).  Tensor form with unpadded rows; results against the oracle, prediction finite."""
    import torch
    from oracle import qfa_oracle as O
    N = 48
    m, rb0, b, p, mu = resident_set(dev, npix, nh, N, seed=7 * npix + nh, stride=npix, with_zabs=zform)
    rb = rb0.with_rows(torch.arange(N, dtype=torch.int32, device=dev))
    rb.error[:, npix - 1] = 0.0
    b["error"][:, npix - 1] = 0.0
    if masked:
        rb.mask[:, npix - 1] = False
        b["mask"][:, npix - 1] = False
    (d, e, z, mk), zfac = rb.materialize()
    m.deterministic = True
    nll = torch.empty(N, dtype=torch.float32, device=dev)
    acc = m.accumulate(d, e, z, mk, zfac=zfac, nll=nll).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(nll).all() and torch.isfinite(acc).all()
    oloss, og = O.forward(p, b["delta"], b["error"], b["zabs"], b["mask"])
    loss, g = m._finalize(acc, True)
    assert abs(loss.item() - oloss) / abs(oloss) < 5e-6
    for key, tol in (("Psi", 2e-5), ("F", 1e-4)):
        ref = np.asarray(og[key])
        ok = ~np.isnan(ref)
        assert rel_l2(g[key].cpu().numpy()[ok], ref[ok]) < tol, key
    (fx, e2, z2, mk2), zfac2 = rb.materialize(raw_flux=True)
    for a in m.predict(fx, e2, z2, mk2, zfac=zfac2):
        assert torch.isfinite(a).all()


@pytest.mark.parametrize("npix,nh,flags", [(1913, 8, 0), (1913, 8, _lib.F_PASS2_PIXRES), (1000, 16, _lib.F_PASS2_PIXRES), (450, 24, 0)])
def test_padded_and_unpadded_rows_give_the_same_bits(dev, npix, nh, flags):
    """row_stride = N_pix (the reference's layout) against rows padded to 32 pixels: identical results"""
    import torch
    N, B = 300, 200
    rows = perm_rows(dev, N, B, seed=5)
    accs = []
    for stride in (npix, (npix + 31) // 32 * 32, npix + 7):
        m, rb, *_ = resident_set(dev, npix, nh, N, seed=11, stride=stride)
        m.flags, m.deterministic = flags, True
        accs.append(m.accumulate(batch=rb.with_rows(rows)).clone())
    assert torch.equal(accs[0], accs[1]) and torch.equal(accs[0], accs[2])


def test_rows_beyond_a_4_gib_offset(dev):
    """the kernels form 64-bit row addresses: rows whose byte offset does not fit 32 bits (a 4.8-GB float array) behave like
    any other row"""
    import torch
    from qfa_amd.resident import ResidentBatch
    npix, nh, B = 4000, 16, 96
    m, rb_small, b, p, mu = resident_set(dev, npix, nh, B, seed=77)
    N = 300_000                                                             # row 299 999 starts at byte 4.8e9
    big = lambda dt, fill: torch.full((N, npix), fill, dtype=dt, device=dev)
    flux, delta, error, mask = big(torch.float32, 0.), big(torch.float32, 0.), big(torch.float32, 1.), big(torch.bool, False)
    zq1 = torch.ones(N, dtype=torch.float32, device=dev)
    where = torch.as_tensor(np.linspace(268_500, N - 1, B).astype(np.int64), device=dev)       # all beyond 2^32 bytes
    assert int(where.min()) * npix * 4 > 2 ** 32
    for dst, src in ((flux, rb_small.flux), (delta, rb_small.delta), (error, rb_small.error), (mask, rb_small.mask)):
        dst[where] = src[:, :npix]
    zq1[where] = rb_small.zq1
    rb_big = ResidentBatch(flux, delta, error, mask, zq1, rb_small.pix_ratio, where.to(torch.int32), npix, rb_small.Nb)
    rb_ref = rb_small.with_rows(torch.arange(B, dtype=torch.int32, device=dev))
    for flags in (0, _lib.F_PASS2_PIXRES):
        m.flags, m.deterministic = flags, True
        assert torch.equal(m.accumulate(batch=rb_big).clone(), m.accumulate(batch=rb_ref).clone())
    m.flags = 0
    for a, c in zip(m.predict(batch=rb_big), m.predict(batch=rb_ref)):
        assert torch.equal(a, c)


def test_argument_checks_of_the_indexed_form(dev):
    import ctypes as C
    import torch
    m, rb, *_ = resident_set(dev, 200, 8, 40, seed=1)
    rb = rb.with_rows(perm_rows(dev, 40, 16, seed=2))
    bs, keep = m._batch_struct_rows(rb)
    ps = m._params_struct()
    ws, acc = m._workspace(16), m._accum()
    call = lambda: _lib.lib().qfa_nll_grad_ex_f32(C.byref(ps), C.byref(bs), C.byref(m._tau_model), 16, m.Npix, m.Nb, m.Nh, None,
                                                  C.c_void_p(acc.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), None, 0, 0,
                                                  _lib.current_stream(dev), None)
    assert call() == 0
    bs.row_stride = m.Npix - 1
    assert call() == -2                                                     # QFA_E_SIZE: rows shorter than N_pix
    bs.row_stride = rb.stride
    bs.A_blue = bs.delta
    assert call() == -1                                                     # QFA_E_NULL: rows are not combined with A_blue
    with pytest.raises(_lib.QFAHipError):
        m._batch_struct_rows(rb.with_rows(rb.rows.long()))                  # int64 rows
    torch.cuda.synchronize()


def _mock_loader(dev, npix, nh, N, B, seed, **kw):
    from qfa_amd import synthetic
    from qfa_amd.dataloader import DeviceDataloader
    wav, nb, nr = synthetic.wavelength_grid(npix)
    p, mu = synthetic.mock_parameters(npix, nb, nh, seed=seed)
    b = synthetic.make_batch_numpy(p, mu, wav, nb, N, seed=seed + 1)
    dl = DeviceDataloader(b["flux"], b["error"], b["zqso"], wav, B, dev, tau="becker", **kw)
    return dl, p, wav, nb, nr


@pytest.mark.parametrize("sort_batches", [True, False])
def test_loader_rows_batches_equal_its_materialised_batches(dev, sort_batches):
    """DeviceDataloader.next_batch_rows() against next_batch() over a shuffled epoch: the same rows in the same order; delta /
    error / mask of the resident arrays equal the per-batch builder's bit for bit (qfa_build_resident_f32 and
    qfa_build_batch_f32 share their arithmetic); the last, short batch included; set_tau rebuilds the resident delta"""
    import torch
    dl, p, wav, nb, nr = _mock_loader(dev, 1913, 8, 150, 64, seed=21)
    dl2 = _mock_loader(dev, 1913, 8, 150, 64, seed=21)[0]                   # the same data walked through next_batch()
    dl.sort_batches = dl2.sort_batches = sort_batches
    assert dl._stride == 1920
    for epoch in range(2):
        np.random.seed(100 + epoch)
        dl.rewind()
        np.random.seed(100 + epoch)
        dl2.rewind()
        refs = []
        while dl.have_next_batch():
            rb = dl.next_batch_rows()
            refs.append(rb)
            d, e, z, mk = dl2.next_batch()
            (d2, e2, _, mk2), zfac = rb.materialize()
            if sort_batches:                                                # the same SET of spectra, rows in storage order
                assert torch.equal(rb.rows, torch.sort(rb.rows)[0])
                back = torch.as_tensor(np.argsort(np.argsort(dl2._order[dl2.cur - d.shape[0]:dl2.cur])), device=dev)
                d2, e2, mk2, zq = d2[back], e2[back], mk2[back], zfac[0][back]
            else:
                zq = zfac[0]
            assert torch.equal(d, d2) and torch.equal(e, e2) and torch.equal(mk, mk2)
            assert torch.equal(z.zfac[0], zq)
        assert not dl2.have_next_batch()
        assert [r.B for r in refs] == [64, 64, 22]
    seen = torch.cat([r.rows for r in refs]).cpu().numpy()
    assert sorted(seen.tolist()) == list(range(150))
    before = dl._delta_pad.clone()
    dl.set_tau("kamble")
    assert not torch.equal(before, dl._delta_pad)
    d, e, z, mk = dl._build(np.arange(150))
    assert torch.equal(dl._delta_pad[:, :1913], d)


class _ReferenceContractOnly(object):
    """a loader that offers the reference's contract and nothing else (QFA/dataloader.py:114-138,154-167): QFA.train then
    walks materialised batches"""

    def __init__(self, dl):
        self._dl = dl
        self.batch_size, self.data_size = dl.batch_size, dl.data_size

    mu = property(lambda self: self._dl.mu)

    def rewind(self):
        self._dl.rewind()

    def have_next_batch(self):
        return self._dl.have_next_batch()

    def next_batch(self):
        return self._dl.next_batch()


@pytest.mark.parametrize("nh,use_graph", [(8, False), (8, 1), (8, 2), (8, 3), (12, False)])
def test_train_through_the_resident_form_reproduces_the_materialised_loop(dev, nh, use_graph, tmp_path):
    """QFA.train takes the resident form of a DeviceDataloader by itself (next_batch is never called); against the same
    loop over materialised batches with the same seeds: identical parameters (deterministic accumulation).  use_graph = k:
    the step graph replays k consecutive steps per launch (5 batches per epoch: k = 2 and 3 leave an eager tail)"""
    import torch
    from qfa_amd import QFA, Adam, step_scheduler
    npix, N, B = 1913, 320, 64
    res = []
    for mode in ("resident", "materialised"):
        dl, p, wav, nb, nr = _mock_loader(dev, npix, nh, N, B, seed=33)
        dl.sort_batches = False                             # (bit comparison: the same summation order in both loops)
        m = QFA(nb, nr, nh, dev, model_params=p)
        m.deterministic = True
        opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 2), learning_rate=2e-3)
        np.random.seed(7)
        if mode == "resident":
            def refuse(*a, **k):
                raise AssertionError("train() materialised a batch of a resident loader")
            dl.next_batch = refuse
            loader = dl
        else:
            loader = _ReferenceContractOnly(dl)
        # (the reference loop is always eager: a step graph fed by a foreign loader copies its batches into fixed buffers and
        # so loses the factored-z attribute of the zabs tensor -- same numbers to rounding, not to the bit)
        m.train(opt, loader, 3, output_dir=str(tmp_path / mode), save_interval=100, smooth_interval=2, quiet=True,
                use_graph=bool(use_graph) and mode == "resident", graph_steps=int(use_graph) or 1)
        res.append({k: getattr(m, k).clone() for k in ("F", "Psi", "omega", "tau0", "c0", "beta")})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


def test_sorted_batches_train_like_unsorted_ones(dev, tmp_path):
    """sort_batches (the default: the rows of a batch in storage order) changes the order of a sum, nothing else"""
    import torch
    from qfa_amd import QFA, Adam, step_scheduler
    res = []
    for sort in (True, False):
        dl, p, wav, nb, nr = _mock_loader(dev, 640, 12, 300, 64, seed=41)
        dl.sort_batches = sort
        m = QFA(nb, nr, 12, dev, model_params=p)
        opt = Adam(m.parameters, dev, scheduler=step_scheduler(0.9, 2), learning_rate=2e-3)
        np.random.seed(3)
        m.train(opt, dl, 3, output_dir=str(tmp_path / str(sort)), save_interval=100, smooth_interval=100, quiet=True)
        res.append({k: getattr(m, k).double().cpu().numpy() for k in ("F", "Psi", "omega", "tau0", "c0", "beta")})
    for k in res[0]:
        assert rel_l2(res[0][k], res[1][k]) < 2e-5, k


def test_predict_to_npz_reads_the_resident_rows(dev, tmp_path):
    import torch
    dl, p, wav, nb, nr = _mock_loader(dev, 450, 8, 37, 16, seed=9, shuffle=False, mode="predict")
    from qfa_amd import QFA
    m = QFA(nb, nr, 8, dev, model_params=p)
    m.mu = torch.tensor(np.asarray(dl.mu), dtype=torch.float32, device=dev)
    names = m.predict_to_npz(dl, str(tmp_path), batch_size=16)
    assert len(names) == 37
    f, e, z, mk, _ = dl.get_rows(0, 37)
    ll, hm, hc, cont, unc = m.predict(f, e, z, mk)
    for r in (0, 17, 36):
        got = np.load(tmp_path / names[r])
        assert np.array_equal(got["cont"], cont[r].cpu().numpy()) and np.array_equal(got["ll"].ravel(), ll[r:r + 1].cpu().numpy())
