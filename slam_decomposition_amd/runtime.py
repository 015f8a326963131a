"""Process-wide libslamhip contexts: one per GPU, plus numbered extra ones for shards that share a GPU."""
from __future__ import annotations

from typing import Dict

from . import _ffi

_contexts: Dict[tuple, "_ffi.Context"] = {}


def get_context(device: int = 0, slot: int = 0) -> "_ffi.Context":
    """The shared context number ``slot`` of ``device`` (created on first use and kept: a context's stream, pinned staging and work
    buffers cost more to set up than a span loop takes; raises if libslamhip.so or the GPU is missing -- there is no CPU fallback).
    Slot 0 is the one the single-context paths use; target shards that run side by side on one GPU take slots 0, 1, ..."""
    ctx = _contexts.get((device, slot))
    if ctx is None:
        ctx = _ffi.Context(device)
        _contexts[(device, slot)] = ctx
    return ctx


def close_all() -> None:
    for ctx in _contexts.values():
        ctx.close()
    _contexts.clear()
