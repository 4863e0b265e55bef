"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares; the Python surface mirrors the reference's names; nothing computes without a GPU."""
import inspect
import os
import re

import numpy as np
import pytest

from conftest import REPO


def _header_symbols():
    txt = open(os.path.join(REPO, "include", "qfa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qfa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from qfa_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    h = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(h, s), s
    assert sorted(_lib.EXPORTS) == syms
    assert _lib.lib().qfa_abi_version() == _lib.ABI_VERSION == 4


def test_host_only_entry_points():
    """qfa_tau_model / qfa_workspace_bytes / qfa_accum_floats are pure host functions."""
    import ctypes as C
    from oracle import qfa_oracle as O
    from qfa_amd import _lib
    for which in ("becker", "fg", "kamble", "mock"):
        for series in (1, 2, 30):
            t = _lib.tau_model(which, series)
            amp, scale, expo, off = O.TAU_MODELS[which]
            c = O.LYMAN_COEFF[series - 1]
            assert abs(t.amp - amp * c) <= 1e-7 * abs(amp * c) + 1e-12
            assert abs(t.offset - off * c) <= 1e-7 * abs(off * c) + 1e-12
            assert abs(t.scale - scale) < 1e-7 and abs(t.expo - expo) < 1e-6
    with pytest.raises(NotImplementedError):
        _lib.tau_model("nope")
    h = _lib.lib()
    assert h.qfa_tau_model(0, 31, C.byref(_lib.TauModel())) == -4
    assert h.qfa_accum_floats(1913, 720, 8) == 1913 * 8 + 3 * 1913 + 720 + 8
    assert h.qfa_workspace_bytes(128, 1913, 8) > 0
    assert h.qfa_workspace_bytes(0, 1913, 8) == 0
    assert h.qfa_workspace_bytes(4, 100, 33) == 0 and h.qfa_workspace_bytes(4, 100, 32) > 0
    # argument validation happens before any device work
    assert h.qfa_nll_grad_f32(None, None, None, 1, 1, 1, 1, None, None, None, 0, None) == -1
    assert h.qfa_adam_clip_f32(None, None, None, None, None, 4, 0, 0, 0, 0, 0, 0, 0, 0, None) == -1


def test_python_surface_matches_reference_names():
    from qfa_amd import model, optimizer, utils
    sig = inspect.signature(model.QFA.__init__)
    assert list(sig.parameters)[1:] == ["Nb", "Nr", "Nh", "device", "tau", "model_params"]
    for name in ("forward", "loglikelihood_and_gradient_for_single_spectra", "prediction_for_single_spectra",
                 "train", "clip", "smooth", "save_to_npz", "load_from_npz", "random_init_func", "parameters",
                 "fit", "predict"):
        assert hasattr(model.QFA, name), name
    assert list(inspect.signature(model.QFA.forward).parameters)[1:5] == ["delta", "error", "zabs", "mask"]
    # the reference's parameters first and in its order; extensions (hipGraph replay) only behind them
    assert list(inspect.signature(model.QFA.train).parameters)[1:9] == [
        "optimizer", "dataloader", "n_epochs", "output_dir", "save_interval", "smooth_interval", "quiet", "logger"]
    assert inspect.signature(model.QFA.train).parameters["use_graph"].default is False
    assert model.QFAModel is model.QFA
    a = inspect.signature(optimizer.Adam.__init__)
    assert list(a.parameters)[1:] == ["params", "device", "scheduler", "learning_rate", "b1", "b2", "eps",
                                      "weight_decay"]
    assert a.parameters["learning_rate"].default == 1e-2 and a.parameters["weight_decay"].default == 1e-3
    for name in ("update", "step", "reset", "scheduled_lr"):
        assert hasattr(optimizer.Adam, name)
    assert optimizer.step_scheduler(0.9, 2)(3, 1.0) == pytest.approx(0.81)
    for name in ("MatrixInverse", "MatrixLogDet", "tauHI", "omega_func", "tau"):
        assert hasattr(utils, name)
    assert model.log2pi == 1.8378770664093453


def test_no_cpu_fallback():
    import torch
    from qfa_amd import QFA, utils
    from qfa_amd._lib import QFAHipError
    with pytest.raises(QFAHipError):
        QFA(4, 4, 2, torch.device("cpu"))
    with pytest.raises(QFAHipError):
        utils.tau(torch.zeros(4))
    with pytest.raises(QFAHipError):
        utils.MatrixLogDet(torch.zeros(4, 2), torch.ones(4))


def test_product_package_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "qfa_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f


def test_synthetic_generator_is_deterministic_and_masks_hold_sentinels():
    from qfa_amd import synthetic
    wav, nb, nr = synthetic.wavelength_grid()
    assert (len(wav), nb, nr) == (1913, 720, 1193)
    for n, enb in ((2000, 753), (4000, 1506), (8000, 3011)):
        assert synthetic.wavelength_grid(n)[1] == enb
    p, mu = synthetic.mock_parameters(400, synthetic.wavelength_grid(400)[1], 8, seed=1)
    w, b_, _ = synthetic.wavelength_grid(400)
    a = synthetic.make_batch_numpy(p, mu, w, b_, 5, seed=9)
    b = synthetic.make_batch_numpy(p, mu, w, b_, 5, seed=9)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert (a["flux"][~a["mask"]] == -999).all() and (a["error"][~a["mask"]] == -999).all()
