"""ctypes binding of libqfa_hip.so (C-ABI declared in include/qfa_hip.h).

There is deliberately no fallback: if the shared library is missing, cannot be loaded, or a
tensor is not a contiguous float32/bool tensor on a HIP device, a ``QFAHipError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (no environment switch: tools/with_lib.py sets this attribute before the first lib() call for A/B timing of library variants)
LIB_PATH = os.path.join(_HERE, "libqfa_hip.so")

EXPORTS = (
    "qfa_abi_version", "qfa_tau_model", "qfa_workspace_bytes", "qfa_accum_floats",
    "qfa_nll_grad_f32", "qfa_nll_grad_events_f32", "qfa_nll_grad_det_f32", "qfa_nll_grad_ex_f32", "qfa_predict_ex_f32", "qfa_det_slab_bytes", "qfa_finalize_grads_f32", "qfa_predict_f32", "qfa_predict_events_f32",
    "qfa_adam_clip_f32",
    "qfa_adam_clip_multi_f32", "qfa_clip_f32", "qfa_smooth_f32", "qfa_tau_f32", "qfa_tauhi_f32", "qfa_omega_func_f32", "qfa_woodbury_f32", "qfa_build_batch_f32", "qfa_mu_estimate_f64",
    "qfa_mu_sums_f64", "qfa_mu_finish_f64", "qfa_build_resident_f32", "qfa_finalize_adam_clip_f32", "qfa_zabs_factor_f32",
)

TAU_IDS = {"becker": 0, "fg": 1, "kamble": 2, "mock": 3}
ABI_VERSION = 4
# `flags` of qfa_nll_grad_ex_f32 / qfa_predict_ex_f32 (include/qfa_hip.h QFA_F_*)
F_PASS2_F32, F_PASS2_XDL, F_S3_FAST, F_PREDICT_F32, F_SYNC = 0x1, 0x2, 0x4, 0x8, 0x20
F_PASS2_PIXRES = 0x40
F_ZERO_ACCUM = 0x80


class QFAHipError(RuntimeError):
    pass


class TauModel(C.Structure):
    _fields_ = [("amp", C.c_float), ("scale", C.c_float), ("expo", C.c_float), ("offset", C.c_float)]


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("F", "Psi", "omega", "tau0", "c0", "beta")]


ADAM_MAX = 8


class AdamMulti(C.Structure):       # qfa_adam_multi_t
    _fields_ = [("p", C.c_void_p * ADAM_MAX), ("g", C.c_void_p * ADAM_MAX), ("m", C.c_void_p * ADAM_MAX),
                ("v", C.c_void_p * ADAM_MAX), ("p_out", C.c_void_p * ADAM_MAX), ("n", C.c_size_t * ADAM_MAX),
                ("lo", C.c_float * ADAM_MAX), ("hi", C.c_float * ADAM_MAX), ("count", C.c_int)]


class Batch(C.Structure):
    # qfa_batch_t (ABI v3: rows / row_stride = the resident, indexed input form)
    _fields_ = [(n, C.c_void_p) for n in ("delta", "error", "zabs", "mask", "A_blue", "zq1", "pix_ratio", "rows")] + \
               [("row_stride", C.c_int64)]


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises QFAHipError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QFAHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C qfa_amd/csrc` (hipcc --offload-arch=gfx950). qfa_amd has no CPU fallback.")
    try:
        h = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise QFAHipError(f"cannot load {LIB_PATH}: {e}") from e
    p, i, f, d, sz, i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_int64
    sigs = {
        "qfa_abi_version": (i, []),
        "qfa_tau_model": (i, [i, i, C.POINTER(TauModel)]),
        "qfa_workspace_bytes": (sz, [i, i, i]),
        "qfa_accum_floats": (sz, [i, i, i]),
        "qfa_nll_grad_f32": (i, [C.POINTER(Params), C.POINTER(Batch), C.POINTER(TauModel), i, i, i, i, p, p, p, sz, p]),
        "qfa_nll_grad_events_f32": (i, [C.POINTER(Params), C.POINTER(Batch), C.POINTER(TauModel), i, i, i, i, p, p, p, sz,
                                        p, C.POINTER(C.c_void_p)]),
        "qfa_nll_grad_det_f32": (i, [C.POINTER(Params), C.POINTER(Batch), C.POINTER(TauModel), i, i, i, i, p, p, p, sz,
                                     p, sz, p, C.POINTER(C.c_void_p)]),
        "qfa_nll_grad_ex_f32": (i, [C.POINTER(Params), C.POINTER(Batch), C.POINTER(TauModel), i, i, i, i, p, p, p, sz,
                                    p, sz, C.c_uint, p, C.POINTER(C.c_void_p)]),
        "qfa_predict_ex_f32": (i, [C.POINTER(Params), p, C.POINTER(Batch), C.POINTER(TauModel), i, i, i, i,
                                   p, p, p, p, p, p, sz, C.c_uint, p, C.POINTER(C.c_void_p)]),
        "qfa_det_slab_bytes": (sz, [i, i, i, i]),
        "qfa_finalize_grads_f32": (i, [p, p, i, i, i, i, p, p, p, p, p, p, p, p]),
        "qfa_predict_f32": (i, [C.POINTER(Params), p, C.POINTER(Batch), C.POINTER(TauModel), i, i, i, i,
                                p, p, p, p, p, p, sz, p]),
        "qfa_predict_events_f32": (i, [C.POINTER(Params), p, C.POINTER(Batch), C.POINTER(TauModel), i, i, i, i,
                                       p, p, p, p, p, p, sz, p, C.POINTER(C.c_void_p)]),
        "qfa_adam_clip_f32": (i, [p, p, p, p, p, sz, d, d, d, d, d, i, f, f, p]),
        "qfa_adam_clip_multi_f32": (i, [C.POINTER(AdamMulti), d, d, d, d, d, i, p]),
        "qfa_finalize_adam_clip_f32": (i, [p, i, i, i, C.POINTER(AdamMulti), d, d, d, d, d, i, p, p]),
        "qfa_clip_f32": (i, [p, p, sz, f, f, p]),
        "qfa_smooth_f32": (i, [p, p, i, i, i, p]),
        "qfa_tau_f32": (i, [p, p, sz, C.POINTER(TauModel), p]),
        "qfa_tauhi_f32": (i, [p, p, p, p, sz, p]),
        "qfa_omega_func_f32": (i, [p, p, p, p, p, sz, p]),
        "qfa_woodbury_f32": (i, [p, p, i, i, p, p, p, sz, p]),
        "qfa_build_batch_f32": (i, [p, p, p, p, p, d, p, i, i, i, i, i64, p, p, p, p, p]),
        "qfa_build_resident_f32": (i, [p, p, p, p, d, p, i, i64, i, i, i64, p, p, p, p]),
        "qfa_mu_estimate_f64": (i, [p, p, p, p, d, i, i, i, i, i64, i, p, p, p, p]),
        "qfa_mu_sums_f64": (i, [p, p, p, p, d, i, i, i, i, i64, p, p]),
        "qfa_mu_finish_f64": (i, [p, i, i, p, p, p]),
        "qfa_zabs_factor_f32": (i, [p, i, i, f, p, p, p, p]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(h, name, None)
        if fn is None:
            raise QFAHipError(f"{LIB_PATH} does not export {name} (include/qfa_hip.h): rebuild it")
        fn.restype = res
        fn.argtypes = args
    if h.qfa_abi_version() != ABI_VERSION:
        raise QFAHipError(f"libqfa_hip.so ABI version {h.qfa_abi_version()}, expected {ABI_VERSION}")
    _lib = h
    return h


def check(status, what):
    if status == 0:
        return
    if status < 0:
        names = {-1: "QFA_E_NULL", -2: "QFA_E_SIZE", -3: "QFA_E_WORKSPACE", -4: "QFA_E_TAU", -5: "QFA_E_FLAGS"}
        raise QFAHipError(f"{what}: invalid argument ({names.get(status, status)})")
    raise QFAHipError(f"{what}: hipError_t {status}")


def tau_model(which="becker", series=1):
    if which not in TAU_IDS:
        raise NotImplementedError("currently available mean optical depth function: ['becker', 'fg', 'kamble']")
    t = TauModel()
    check(lib().qfa_tau_model(TAU_IDS[which], int(series), C.byref(t)), "qfa_tau_model")
    return t


def require_device_tensor(t, dtype, name):
    """Raw pointer of a contiguous tensor on a HIP device, or a loud error."""
    import torch
    if not isinstance(t, torch.Tensor):
        raise QFAHipError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if t.device.type != "cuda":
        raise QFAHipError(f"{name}: tensor is on {t.device}; qfa_amd runs on a HIP device only (no CPU fallback)")
    if t.dtype != dtype:
        raise QFAHipError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise QFAHipError(f"{name}: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
