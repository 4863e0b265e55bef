"""Adam with the reference's surface and quirks (reference QFA/optimizer.py:11-99).

* L2 regularisation is folded into the gradient (optimizer.py:47);
* the bias-correction exponent and the learning-rate schedule use ``self.i``, which only
  advances in ``step()`` -- once per epoch in ``QFA.train`` (model.py:215), quirk Q4;
* ``update`` is functional: it returns a new parameter dict and leaves its input untouched.
The arithmetic runs in ``qfa_adam_clip_multi_f32`` (all tensors of the dict in one launch).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Tuple

import torch

from . import _lib

f32 = torch.float32


class Adam(object):

    def __init__(self, params: Dict[str, torch.Tensor], device: torch.device, scheduler=None,
                 learning_rate: float = 1e-2, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8,
                 weight_decay: float = 1e-3) -> None:
        self.learning_rate = learning_rate
        self.b1 = b1
        self.b2 = b2
        self.eps = eps
        self.device = device
        self.weight_decay = weight_decay
        self.scheduler = scheduler
        self.reset(params)

    def reset(self, params):
        """reference QFA/optimizer.py:54-63"""
        self.m = {k: torch.zeros_like(params[k], dtype=f32, device=self.device) for k in params}
        self.v = {k: torch.zeros_like(params[k], dtype=f32, device=self.device) for k in params}
        self.i = 0

    def step(self):
        """reference QFA/optimizer.py:65-69"""
        self.i += 1

    @property
    def scheduled_lr(self):
        """reference QFA/optimizer.py:71-76"""
        if callable(self.scheduler):
            return self.scheduler(self.i, self.learning_rate)
        return self.learning_rate

    def update(self, params, g, clip: Optional[Dict[str, Tuple[float, float]]] = None, inplace: bool = False):
        """reference QFA/optimizer.py:37-52.  ``clip`` (key -> (lo, hi)) optionally fuses the
        clamp of QFA.clip into the same launch; the plain reference call leaves it out and the
        ``QFA.parameters`` setter clips afterwards.  ``inplace`` writes the new values over ``params``
        (fixed addresses: what a captured hipGraph of the step needs) instead of returning new tensors."""
        h = _lib.lib()
        lr = float(self.scheduled_lr)
        out, keep = {}, []
        keys = list(params)
        for c0 in range(0, len(keys), _lib.ADAM_MAX):               # one launch per (up to) 8 tensors
            t = _lib.AdamMulti()
            chunk = keys[c0:c0 + _lib.ADAM_MAX]
            for j, k in enumerate(chunk):
                p = params[k]
                if p.dtype != f32 or not p.is_contiguous():
                    p = p.to(f32).contiguous()
                grad = g[k]
                if grad.dtype != f32 or not grad.is_contiguous():
                    grad = grad.to(f32).contiguous()
                _lib.require_device_tensor(p, f32, f"params[{k}]")
                _lib.require_device_tensor(grad, f32, f"g[{k}]")
                if inplace and p is not params[k]:
                    raise ValueError(f"in-place update needs contiguous float32 params[{k}]")
                q = p if inplace else torch.empty_like(p)
                lo, hi = clip[k] if (clip is not None and k in clip) else (1.0, 0.0)
                t.p[j], t.g[j], t.m[j], t.v[j], t.p_out[j] = (p.data_ptr(), grad.data_ptr(), self.m[k].data_ptr(),
                                                              self.v[k].data_ptr(), q.data_ptr())
                t.n[j], t.lo[j], t.hi[j] = p.numel(), lo, hi
                keep += [p, grad]
                out[k] = q
            t.count = len(chunk)
            dev = params[chunk[0]].device
            _lib.check(h.qfa_adam_clip_multi_f32(C.byref(t), lr, self.b1, self.b2, self.eps, self.weight_decay,
                                                 int(self.i), _lib.current_stream(dev)), "qfa_adam_clip_multi_f32")
        return out

    def update_from_accum(self, model, acc, clip=None, inplace=False):
        """``forward``'s normalisation and ``update`` in ONE launch (qfa_finalize_adam_clip_f32): the packed, all-reduced sum /
        count buffer ``acc`` of the step goes straight to the new parameters; the gradients are never materialised.
        Returns (loss (1,1), new parameter dict); bit-identical to ``update(params, model._finalize(acc)[1], ...)``."""
        keys = ("F", "Psi", "omega", "tau0", "c0", "beta")
        params = model.parameters
        t = _lib.AdamMulti()
        out, keep = {}, []
        for j, k in enumerate(keys):
            p = params[k]
            if p.dtype != f32 or not p.is_contiguous():
                if inplace:
                    raise ValueError(f"in-place update needs contiguous float32 params[{k}]")
                p = p.to(f32).contiguous()
            _lib.require_device_tensor(p, f32, f"params[{k}]")
            q = p if inplace else torch.empty_like(p)
            lo, hi = clip[k] if (clip is not None and k in clip) else (1.0, 0.0)
            t.p[j], t.g[j], t.m[j], t.v[j], t.p_out[j] = p.data_ptr(), None, self.m[k].data_ptr(), self.v[k].data_ptr(), q.data_ptr()
            t.n[j], t.lo[j], t.hi[j] = p.numel(), lo, hi
            keep.append(p)
            out[k] = q
        t.count = 6
        loss = torch.empty((1, 1), dtype=f32, device=model.device)
        _lib.check(_lib.lib().qfa_finalize_adam_clip_f32(
            C.c_void_p(acc.data_ptr()), model.Npix, model.Nb, model.Nh, C.byref(t), float(self.scheduled_lr), self.b1, self.b2,
            self.eps, self.weight_decay, int(self.i), C.c_void_p(loss.data_ptr()), _lib.current_stream(model.device)),
            "qfa_finalize_adam_clip_f32")
        return loss, out

    # checkpoint support for the optimiser state (absent in the reference)
    def state_dict(self):
        return {"i": self.i, "m": {k: v.clone() for k, v in self.m.items()},
                "v": {k: v.clone() for k, v in self.v.items()}}

    def load_state_dict(self, sd):
        self.i = int(sd["i"])
        self.m = {k: v.to(self.device, f32).clone() for k, v in sd["m"].items()}
        self.v = {k: v.to(self.device, f32).clone() for k, v in sd["v"].items()}


def step_scheduler(alpha: float, step: int) -> Callable[[int, float], float]:
    """lr * alpha ** ((i+1)//step)  (reference QFA/optimizer.py:79-99)."""
    def scheduler(i, lr):
        return lr * alpha ** ((i + 1) // step)
    return scheduler
