"""Process-wide libslamhip contexts, one per GPU."""
from __future__ import annotations

from typing import Dict

from . import _ffi

_contexts: Dict[int, "_ffi.Context"] = {}


def get_context(device: int = 0) -> "_ffi.Context":
    """The shared context of ``device`` (created on first use; raises if libslamhip.so or the GPU
    is missing -- there is no CPU fallback)."""
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _ffi.Context(device)
        _contexts[device] = ctx
    return ctx


def close_all() -> None:
    for ctx in _contexts.values():
        ctx.close()
    _contexts.clear()
