import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A zabs tensor handed to the model twice is tested for the reference loader's structure and then served by the factored-z
    # kernels (QFA.auto_factor_zabs).  The suites pass zabs tensors to exercise the ZABS kernels: off by default here,
    # tests/test_hip_parity.py::test_auto_factored_zabs_* turn it on.
    import qfa_amd.model as M
    M.AUTO_FACTOR_ZABS = False


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="session")
def shipped():
    """Shipped SDSS parameters through load_from_npz semantics (c0 <- beta quirk on)."""
    from oracle import qfa_oracle as O
    return O.load_params_npz(os.path.join(GOLDEN, "model_parameters.npz"))


@pytest.fixture(scope="session")
def grid():
    from qfa_amd import synthetic
    return synthetic.wavelength_grid()
