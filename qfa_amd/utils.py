"""Math helpers with the reference's names and argument meaning (reference QFA/utils.py).

``tau``, ``tauHI``, ``omega_func``, ``MatrixInverse`` and ``MatrixLogDet`` take torch tensors on a
HIP device and run HIP kernels through the C-ABI; CPU tensors raise ``QFAHipError``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

f32 = torch.float32


def _as_f32(x, device=None):
    if not isinstance(x, torch.Tensor):
        raise _lib.QFAHipError(f"expected a torch.Tensor on a HIP device, got {type(x)}")
    if x.dtype != f32 or not x.is_contiguous():
        x = x.to(f32).contiguous()
    return x


def tau(z: torch.Tensor, which: Optional[str] = "becker", series: Optional[int] = 1) -> torch.Tensor:
    """Mean optical depth (reference QFA/utils.py:149-171)."""
    t = _lib.tau_model(which, series)
    z = _as_f32(z)
    zp = _lib.require_device_tensor(z, f32, "z")
    out = torch.empty_like(z)
    _lib.check(_lib.lib().qfa_tau_f32(zp, C.c_void_p(out.data_ptr()), z.numel(), C.byref(t),
                                      _lib.current_stream(z.device)), "qfa_tau_f32")
    return out


def _scalar_on(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=f32).reshape(1).contiguous()
    return torch.tensor([float(x)], dtype=f32, device=device)


def tauHI(z: torch.Tensor, tau0, beta) -> torch.Tensor:
    """tau0 * (1+z)**beta (reference QFA/utils.py:57-73)."""
    z = _as_f32(z)
    zp = _lib.require_device_tensor(z, f32, "z")
    t0, be = _scalar_on(tau0, z.device), _scalar_on(beta, z.device)
    out = torch.empty_like(z)
    _lib.check(_lib.lib().qfa_tauhi_f32(zp, C.c_void_p(t0.data_ptr()), C.c_void_p(be.data_ptr()),
                                        C.c_void_p(out.data_ptr()), z.numel(), _lib.current_stream(z.device)),
               "qfa_tauhi_f32")
    return out


def omega_func(z: torch.Tensor, tau0, beta, c0) -> torch.Tensor:
    """(1 - c0 - exp(-tau0 (1+z)^beta))^2 (reference QFA/utils.py:76-92)."""
    z = _as_f32(z)
    zp = _lib.require_device_tensor(z, f32, "z")
    t0, be, cc = _scalar_on(tau0, z.device), _scalar_on(beta, z.device), _scalar_on(c0, z.device)
    out = torch.empty_like(z)
    _lib.check(_lib.lib().qfa_omega_func_f32(zp, C.c_void_p(t0.data_ptr()), C.c_void_p(be.data_ptr()),
                                             C.c_void_p(cc.data_ptr()), C.c_void_p(out.data_ptr()), z.numel(),
                                             _lib.current_stream(z.device)), "qfa_omega_func_f32")
    return out


def _woodbury(M, D, want_inv, want_logdet):
    M, D = _as_f32(M), _as_f32(D)
    mp = _lib.require_device_tensor(M, f32, "M")
    dp = _lib.require_device_tensor(D, f32, "D")
    n, k = M.shape
    ws = torch.empty((k * k + 1) * 8, dtype=torch.uint8, device=M.device)
    inv = torch.empty((n, n), dtype=f32, device=M.device) if want_inv else None
    ld = torch.empty((), dtype=f32, device=M.device) if want_logdet else None
    _lib.check(_lib.lib().qfa_woodbury_f32(mp, dp, n, k, C.c_void_p(inv.data_ptr() if want_inv else None),
                                           C.c_void_p(ld.data_ptr() if want_logdet else None),
                                           C.c_void_p(ws.data_ptr()), ws.numel(), _lib.current_stream(M.device)),
               "qfa_woodbury_f32")
    return inv, ld


def MatrixInverse(M: torch.Tensor, D: torch.Tensor, device: torch.device = None) -> torch.Tensor:
    """Dense inverse of M M^T + diag(D) (reference QFA/utils.py:12-32)."""
    return _woodbury(M, D, True, False)[0]


def MatrixLogDet(M: torch.Tensor, D: torch.Tensor, device: torch.device = None) -> torch.Tensor:
    """log det(M M^T + diag(D)) (reference QFA/utils.py:35-54); finite where float32 det overflows."""
    return _woodbury(M, D, False, True)[1]
