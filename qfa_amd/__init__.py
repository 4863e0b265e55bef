"""qfa_amd -- MI355X-native implementation of the QFA hot path.

The public surface mirrors the reference package (``QFA.model.QFA``, ``QFA.optimizer.Adam``,
``QFA.optimizer.step_scheduler``, ``QFA.utils``); all arithmetic runs in hand-written HIP
kernels for gfx950 behind the C-ABI library ``libqfa_hip.so`` (see ``include/qfa_hip.h``).
Importing this package does not touch the GPU; the library is loaded on first use and
a missing library or device raises ``QFAHipError`` (there is no CPU fallback).
"""
__version__ = "0.1.0"

_LAZY = {
    "QFA": ("qfa_amd.model", "QFA"),
    "QFAModel": ("qfa_amd.model", "QFAModel"),
    "Adam": ("qfa_amd.optimizer", "Adam"),
    "step_scheduler": ("qfa_amd.optimizer", "step_scheduler"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(mod), attr)
    raise AttributeError(name)
