"""The hand-counted waits of the XDL kernels against a build without any.

k_moments_x, k_grads_x and k_predict_x issue their LDS-DMA and (pass 1) their spectra loads as asm statements that hipcc's
s_waitcnt bookkeeping does not see, and retire them with counted `s_waitcnt vmcnt(N)`.  `make -C qfa_amd/csrc tracked`
builds libqfa_tracked.so from the same sources with QFA_TRACKED_LOADS=1: builtin LDS-DMA, ordinary loads, vmcnt(0) and
__syncthreads() at every hand-over -- the compiler keeps the books.  The arithmetic is the same instruction for
instruction, so the two builds must agree BIT FOR BIT; a counted wait that is one request short shows up here as a
difference.  The deterministic mode is used for the gradients (float atomics make the default mode's last bits depend
on timing in either build); the per-spectrum NLL and the prediction outputs have no atomics in them.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACKED = os.path.join(REPO, "qfa_amd", "libqfa_tracked.so")


def run(lib, out, npix, nh, B, form="zabs", pixres=False):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "dump_hot_path.py"), out, str(npix), str(nh), str(B),
                        "deterministic"] + (["zfac"] if form == "zfac" else []) + (["pixres"] if pixres else [])
                       + ([f"lib={lib}"] if lib else []), capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("npix,nh,B", [(4000, 16, 3000),      # k_moments_x<16>, k_grads_x, k_predict_x<16>; full tiles
                                       (1913, 8, 2500),       # k_moments_x<8>, k_predict_x<8>; ragged last tile
                                       (1000, 12, 700),       # N_h = 12 on the 16-wide XDL kernels
                                       (1100, 24, 300),       # N_h = 24: k_moments_x<32> (two column sweeps per tile),
                                                              # k_s12_x, k_grads_s3, k_predict_x32 (tracked qfa_k32 object)
                                       (2050, 32, 1000)])     # N_h = 32, several work items per block, ragged last tile
@pytest.mark.parametrize("form", ["zabs", "zfac"])          # zfac: the factored-z input form (other request counts)
def test_tracked_build_is_bit_identical(tmp_path, npix, nh, B, form):
    assert os.path.exists(TRACKED), "libqfa_tracked.so missing: __graft_entry__.build() / make -C qfa_amd/csrc tracked"
    a = run(None, str(tmp_path / "shipped.npz"), npix, nh, B, form)
    b = run(TRACKED, str(tmp_path / "tracked.npz"), npix, nh, B, form)
    for k in a.files:
        assert np.array_equal(a[k], b[k], equal_nan=True), (k, float(np.nanmax(np.abs(a[k] - b[k]))))


@pytest.mark.parametrize("npix,nh,B", [(4000, 16, 3000),      # k_grads_t: full tiles, several ranges of spectra groups
                                       (1913, 13, 2500),      # ragged last tile (4-byte pieces, full waits), blue/red boundary inside a tile
                                       (1000, 12, 70),        # few groups per range: the first / last steps only
                                       (1913, 8, 2500)])      # k_grads_t<8>
@pytest.mark.parametrize("form", ["zabs", "zfac"])
def test_tracked_build_is_bit_identical_pixel_resident_pass2(tmp_path, npix, nh, B, form):
    """The same comparison for the pixel-resident form of pass 2 (k_grads_t, qfa_grads_t.h: state parts and spectra by
    untracked LDS-DMA, one counted wait per group)."""
    assert os.path.exists(TRACKED), "libqfa_tracked.so missing: __graft_entry__.build() / make -C qfa_amd/csrc tracked"
    a = run(None, str(tmp_path / "shipped.npz"), npix, nh, B, form, pixres=True)
    b = run(TRACKED, str(tmp_path / "tracked.npz"), npix, nh, B, form, pixres=True)
    for k in a.files:
        assert np.array_equal(a[k], b[k], equal_nan=True), (k, float(np.nanmax(np.abs(a[k] - b[k]))))
