"""ctypes binding of ``libslamhip.so`` (C ABI: ``include/slam_hip.h``).

No torch, no NumPy-on-CPU fallback: if the shared library is missing or no GPU
is usable every entry point raises -- the product path never computes on the
host.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLAM_HIP_LIB selects another build of the same library (kernel A/B experiments); never a fallback
LIB_PATH = os.environ.get("SLAM_HIP_LIB") or os.path.join(_HERE, "lib", "libslamhip.so")

ABI_VERSION = 7  # include/slam_hip.h: SLAM_ABI_VERSION
MAX_SPAN_QUAD = 5  # SLAM_MAX_SPAN_QUAD: the register-resident kernels
MAX_SPAN_EVAL = 16
MAX_SPAN_MINIMIZE = 16

ST_CONVERGED, ST_MAXITER, ST_LINESEARCH, ST_NONFINITE, ST_STALLED, ST_PREEMPTED = range(6)
FLAG_EARLY_EXIT = 1
FLAG_ORDERED = 2  # with EARLY_EXIT: the lowest-index successful restart wins (reference semantics, reproducible)
FLAG_STAGED = 4  # span loops: never use the one-wavefront-per-target kernel (small batches)
FLAG_OVERLAP = 8  # SLAM_FLAG_OVERLAP: the spans of a loop side by side whatever the call's size
FLAG_NO_OVERLAP = 16  # SLAM_FLAG_NO_OVERLAP: not even for medium calls (several calls in flight on the device)
FLAG_NO_EXTERIOR = 32  # SLAM_FLAG_NO_EXTERIOR: CircuitTemplate(no_exterior_1q=True) -- layers 0 and k pinned at the identity
MAX_MAXITER = 4000
V2_MAX_SPAN = 5
OP_SUM, OP_MAX, OP_MIN = 0, 2, 3
COMM_ID_BYTES = 128
COST_BASIC, COST_SQUARE = 0, 1

# every symbol include/slam_hip.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = (
    "slam_last_error",
    "slam_version",
    "slam_abi_version",
    "slam_device_count",
    "slam_ctx_create",
    "slam_ctx_destroy",
    "slam_ctx_device_info",
    "slam_set_targets",
    "slam_set_gates",
    "slam_c1c2c3",
    "slam_targets_c1c2c3",
    "slam_predict_spans",
    "slam_eval_c1c2c3",
    "slam_sample_haar",
    "slam_get_targets",
    "slam_eval_loss_grad",
    "slam_eval_unitary",
    "slam_minimize_stage",
    "slam_decompose",
    "slam_decompose_resident",
    "slam_fetch_results",
    "slam_decompose_range",
    "slam_decompose_list",
    "slam_decompose_predicted",
    "slam_decompose_multi",
    "slam_decompose_range_fetch",
    "slam_fetch_results_range",
    "slam_fetch_span_losses",
    "slam_minimize_stage_trace",
    "slam_v2_set_gates",
    "slam_v2_set_constraint",
    "slam_v2_eval_loss_grad",
    "slam_v2_minimize_stage",
    "slam_v2_minimize_stage_trace",
    "slam_v2_decompose_range",
    "slam_set_cost",
    "slam_synchronize",
    "slam_host_alloc",
    "slam_host_free",
    "slam_get_stats",
    "slam_reset_stats",
    "slam_best_loss_device_ptr",
    "slam_ctx_device",
    "slam_comm_get_unique_id",
    "slam_comm_init",
    "slam_comm_destroy",
    "slam_comm_rank",
    "slam_comm_allreduce_f64",
    "slam_comm_barrier",
    "slam_comm_merge_begin",
    "slam_comm_merge_add",
    "slam_comm_merge_add_host",
    "slam_allreduce_min",
)


class SlamHipError(RuntimeError):
    """A libslamhip call returned a negative SLAM_ERR_* code."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libslamhip error {code}: {message}")
        self.code = code


class OptParams(C.Structure):
    _fields_ = [
        ("restarts", C.c_int32),
        ("maxiter", C.c_int32),
        ("gtol", C.c_double),
        ("stop_loss", C.c_double),
        ("seed", C.c_uint64),
        ("flags", C.c_uint32),
        ("items_per_quad", C.c_uint32),
        ("gtol_far", C.c_double),
        ("far_loss", C.c_double),
        ("target_base", C.c_int64),
    ]

    def __init__(self, restarts=5, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=0, flags=0, gtol_far=1e-5, far_loss=1e-6,
                 items_per_quad=0, target_base=0):
        super().__init__(int(restarts), int(maxiter), float(gtol), float(stop_loss), int(seed) & 0xFFFFFFFFFFFFFFFF,
                         int(flags), int(items_per_quad), float(gtol_far), float(far_loss), int(target_base))


class V2Gate(C.Structure):
    """``slam_v2_gate``: raw angles (a, phi_c, b, phi_g)[r] = scale[r] * q[sel[r]] + offset[r] of a conversion-gain gate."""

    _fields_ = [("n_params", C.c_int32), ("sel", C.c_int32 * 4), ("scale", C.c_double * 4), ("offset", C.c_double * 4)]

    def __init__(self, n_params, sel, scale, offset):
        super().__init__(int(n_params), (C.c_int32 * 4)(*[int(v) for v in sel]), (C.c_double * 4)(*[float(v) for v in scale]),
                         (C.c_double * 4)(*[float(v) for v in offset]))


class Stats(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_double),
        ("kernel_launches", C.c_int64),
        ("evals", C.c_int64 * (MAX_SPAN_EVAL + 1)),
        ("items", C.c_int64 * (MAX_SPAN_EVAL + 1)),
        ("total_ms", C.c_double),
        ("kernel_ms_span", C.c_double * (MAX_SPAN_EVAL + 1)),
        ("wave_rounds", C.c_int64 * (MAX_SPAN_EVAL + 1)),
        ("evals_accepted", C.c_int64 * (MAX_SPAN_EVAL + 1)),
        ("evals_preempted", C.c_int64 * (MAX_SPAN_EVAL + 1)),
    ]


_lib = None


def load_library() -> C.CDLL:
    """Load libslamhip.so (built in-tree by ``__graft_entry__.build()`` / ``make``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950); "
            "slam_decomposition_amd has no CPU fallback"
        )
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    dp = np.ctypeslib.ndpointer
    lib.slam_last_error.restype = C.c_char_p
    lib.slam_version.restype = C.c_char_p
    variant = "SLAM_HIP_LIB" in os.environ  # an A/B build selected by a dev tool: older entry points are tolerated
    if hasattr(lib, "slam_abi_version"):
        lib.slam_abi_version.restype = C.c_int
        abi = int(lib.slam_abi_version())
    else:
        abi = 0
    if abi != ABI_VERSION and not variant:
        raise ImportError(f"{LIB_PATH}: binary interface revision {abi}, this binding is written for {ABI_VERSION} (include/slam_hip.h: "
                          "SLAM_ABI_VERSION): rebuild the library with `make`")
    lib.slam_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.slam_ctx_create.argtypes = [C.c_int, C.POINTER(P)]
    lib.slam_ctx_destroy.argtypes = [P]
    lib.slam_ctx_device_info.argtypes = [P, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.slam_set_targets.argtypes = [P, P, C.c_int64]
    lib.slam_set_gates.argtypes = [P, P, C.c_int32]
    lib.slam_c1c2c3.argtypes = [P, P, C.c_int64, C.c_int32, P]
    lib.slam_targets_c1c2c3.argtypes = [P, C.c_int64, C.c_int64, C.c_int32, P]
    lib.slam_predict_spans.argtypes = [P, C.c_int64, C.c_int64, C.c_int32, P, P, C.c_double, P]
    lib.slam_eval_c1c2c3.argtypes = [P, C.c_int32, P, P, C.c_int64, C.c_int32, P]
    lib.slam_sample_haar.argtypes = [P, C.c_uint64, C.c_int64, C.c_int64]
    lib.slam_get_targets.argtypes = [P, C.c_int64, C.c_int64, P]
    lib.slam_eval_loss_grad.argtypes = [P, C.c_int, P, P, P, C.c_int64, P, P]
    lib.slam_eval_unitary.argtypes = [P, C.c_int, P, P, P, C.c_int64, P, P]
    lib.slam_minimize_stage.argtypes = [P, C.c_int, P, P, C.c_int64, P, C.POINTER(OptParams)] + [P] * 7
    lib.slam_decompose.argtypes = [P, C.c_int, C.c_int, P, C.POINTER(OptParams), C.c_double, P, P, P]
    lib.slam_decompose_resident.argtypes = [P, C.c_int, C.c_int, P, C.POINTER(OptParams), C.c_double]
    lib.slam_fetch_results.argtypes = [P, C.c_int, P, P, P]
    lib.slam_decompose_range.argtypes = [P, C.c_int64, C.c_int64, C.c_int, C.c_int, P, C.POINTER(OptParams), C.c_double]
    if hasattr(lib, "slam_decompose_range_fetch"):  # (absent from older A/B builds selected with SLAM_HIP_LIB)
        lib.slam_decompose_range_fetch.argtypes = [P, C.c_int64, C.c_int64, C.c_int, C.c_int, P, C.POINTER(OptParams), C.c_double, P, P, P]
    lib.slam_decompose_list.argtypes = [P, P, C.c_int64, C.c_int, C.c_int, C.c_int, P, C.POINTER(OptParams), C.c_double]
    lib.slam_decompose_predicted.argtypes = [P, C.c_int64, C.c_int64, C.c_int, P, P, C.c_double, C.c_int, P, C.POINTER(OptParams), C.c_double, P, P]
    if hasattr(lib, "slam_decompose_multi"):
        lib.slam_decompose_multi.argtypes = [C.POINTER(P), C.c_int32, C.c_int64, C.c_int64, C.c_int, C.c_int, P, C.POINTER(OptParams), C.c_double]
    lib.slam_fetch_results_range.argtypes = [P, C.c_int, C.c_int64, C.c_int64, P, P, P]
    if hasattr(lib, "slam_minimize_stage_trace"):
        lib.slam_fetch_span_losses.argtypes = [P, C.c_int64, C.c_int64, P]
        lib.slam_minimize_stage_trace.argtypes = [P, C.c_int, P, P, C.c_int64, P, C.POINTER(OptParams), C.c_double, C.c_int32] + [P] * 8
    if hasattr(lib, "slam_v2_set_gates"):
        lib.slam_v2_set_gates.argtypes = [P, C.POINTER(V2Gate), C.c_int32]
        if hasattr(lib, "slam_v2_set_constraint"):
            lib.slam_v2_set_constraint.argtypes = [P, C.c_int, P, C.c_int, C.c_double]
        lib.slam_v2_eval_loss_grad.argtypes = [P, C.c_int, P, P, P, C.c_int64, P, P, P]
        lib.slam_v2_minimize_stage.argtypes = [P, C.c_int, P, P, C.c_int64, P, P, P, P, P, C.POINTER(OptParams), C.c_double] + [P] * 7
        if hasattr(lib, "slam_v2_decompose_range"):
            lib.slam_v2_decompose_range.argtypes = [P, C.c_int64, C.c_int64, C.c_int, C.c_int, P, P, P, P, P, C.POINTER(OptParams), C.c_double, P, P, P]
        if hasattr(lib, "slam_v2_minimize_stage_trace"):
            lib.slam_v2_minimize_stage_trace.argtypes = [P, C.c_int, P, P, C.c_int64, P, P, P, P, P, C.POINTER(OptParams), C.c_double, C.c_int32] + [P] * 8
    lib.slam_set_cost.argtypes = [P, C.c_int]
    lib.slam_synchronize.argtypes = [P]
    if hasattr(lib, "slam_host_alloc"):
        lib.slam_host_alloc.argtypes = [C.c_size_t, C.POINTER(P)]
        lib.slam_host_free.argtypes = [P]
    lib.slam_get_stats.argtypes = [P, C.POINTER(Stats)]
    lib.slam_reset_stats.argtypes = [P]
    lib.slam_best_loss_device_ptr.argtypes = [P, C.POINTER(P), C.POINTER(C.c_int64)]
    if hasattr(lib, "slam_ctx_device"):
        lib.slam_ctx_device.argtypes = [P, C.POINTER(C.c_int)]
    if hasattr(lib, "slam_comm_init") and (abi >= 4 or not variant):  # (older variants: slam_allreduce_min had four arguments)
        lib.slam_comm_get_unique_id.argtypes = [P]
        lib.slam_comm_init.argtypes = [C.c_int, C.c_int, C.c_int, P, C.POINTER(P)]
        lib.slam_comm_destroy.argtypes = [P]
        lib.slam_comm_rank.argtypes = [P, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.slam_comm_allreduce_f64.argtypes = [P, P, C.c_int64, C.c_int]
        lib.slam_comm_barrier.argtypes = [P]
        lib.slam_comm_merge_begin.argtypes = [P, C.c_int64]
        lib.slam_comm_merge_add.argtypes = [P, P, C.c_int64, C.c_int64, C.c_int64]
        lib.slam_comm_merge_add_host.argtypes = [P, P, C.c_int64, C.c_int64]
        lib.slam_allreduce_min.argtypes = [P, C.c_double, C.POINTER(C.c_int64), P, C.c_int64]
    for name in EXPORTED_SYMBOLS:
        if "SLAM_HIP_LIB" in os.environ and not hasattr(lib, name):
            continue  # an older A/B build: newer entry points are simply not used
        fn = getattr(lib, name)
        if name not in ("slam_last_error", "slam_version", "slam_abi_version"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        raise SlamHipError(rc, load_library().slam_last_error().decode("utf-8", "replace"))


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------------------
# Result arrays in page-locked memory, recycled (slam_host_alloc, include/slam_hip.h): a result array of a big window (12.6 MB for
# 65 536 x 24 parameters) is a view of a pinned block that goes back to the pool when the last view of it dies -- the device copies
# into it by DMA, and the call's results are never handed back to the C allocator (measured: freeing 12.6 MB arrays that the runtime
# had pinned around a copy stalled the next call's first launches for 13-18 ms).  For ONE blocking call alone on the device
# (``pinned=True``; TemplateOptimizer's single-call path): 18.2 -> 16.9 ms for 65 536 x 32 sqrt(iSWAP).  NOT for several calls in
# flight: the copy into pinned memory is a device-side copy that needs wave slots behind the other calls' persistent kernels, while
# the pageable path's staging runs on the calling host thread beside them (profiles/r5_pinned_results_ab.txt: the driver's command 14.4 -> 14.9 ms
# per step, 327 680 targets through the API 79 -> 82 ms with everything pinned).  Pageable result arrays are recycled the same way
# (``pageable_pool``): with glibc told never to return memory the 327 680-target call went 86-88 -> 79 ms on a box where the default
# allocator had it in its slow mode (tools/r5_malloc_ab.sh); the pool gets that without touching the process's allocator settings.
# ------------------------------------------------------------------------------------------------
class _PoolBlock:
    __slots__ = ("ptr", "cap", "buf", "_pool", "__weakref__")

    def __init__(self, pool, ptr, cap, buf):
        self.ptr, self.cap, self.buf, self._pool = ptr, cap, buf, pool

    @property
    def __array_interface__(self):
        return {"shape": (self.cap,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            self._pool._release(self.ptr, self.cap, self.buf)
        except Exception:  # (interpreter shutdown)
            pass


class ResultPool:
    """Big result arrays as views of recycled blocks: page-locked ones from ``slam_host_alloc`` (``pinned=True``) or ordinary NumPy
    buffers.  A block goes back to the pool when the last view of it dies."""

    MIN_BYTES = 1 << 18           # smaller arrays stay ordinary NumPy arrays (the library stages small fetches itself)
    GRANULE = 1 << 20
    MAX_IDLE_BYTES = 2 << 30      # idle blocks beyond this are given up

    def __init__(self, pinned: bool):
        import threading

        self.pinned = bool(pinned)
        self._lock = threading.RLock()  # (re-entrant: a block's __del__ may run -- garbage collection -- while this thread holds it)
        self._free = {}
        self._idle = 0
        self.allocated = 0        # blocks obtained so far (tests / diagnostics)

    def empty(self, shape, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype=dtype)
        cap = -(-nbytes // self.GRANULE) * self.GRANULE
        entry = None
        with self._lock:
            lst = self._free.get(cap)
            if lst:
                entry = lst.pop()
                self._idle -= cap
        if entry is None:
            if self.pinned:
                lib = load_library()
                out = C.c_void_p()
                if not hasattr(lib, "slam_host_alloc") or lib.slam_host_alloc(cap, C.byref(out)) != 0 or not out.value:
                    return np.empty(shape, dtype=dtype)  # (no page-locked memory: pageable results are slower, not wrong)
                entry = (int(out.value), None)
            else:
                buf = np.empty(cap, dtype=np.uint8)
                entry = (buf.ctypes.data, buf)
            self.allocated += 1
        block = _PoolBlock(self, entry[0], cap, entry[1])
        return np.asarray(block)[:nbytes].view(dtype).reshape(shape)

    def _release(self, ptr, cap, buf):
        with self._lock:
            if self._idle + cap <= self.MAX_IDLE_BYTES:
                self._free.setdefault(cap, []).append((ptr, buf))
                self._idle += cap
                return
        if buf is None:
            load_library().slam_host_free(C.c_void_p(ptr))


PinnedPool = ResultPool  # (name of the first version, tests)
result_pool = ResultPool(pinned=True)      # ONE blocking call alone on the device (``pinned=True`` below)
pageable_pool = ResultPool(pinned=False)   # everything else


def _mat_to_ri(mats: np.ndarray) -> np.ndarray:
    """complex128[..., 4, 4] -> float64[..., 4, 4, 2] (row-major re, im), contiguous."""
    m = np.ascontiguousarray(np.asarray(mats, dtype=np.complex128))
    if m.shape[-2:] != (4, 4):
        raise ValueError(f"expected 4x4 matrices, got shape {m.shape}")
    return m.view(np.float64).reshape(m.shape + (2,))


def device_count() -> int:
    n = C.c_int(0)
    _check(load_library().slam_device_count(C.byref(n)))
    return n.value


class Context:
    """One GPU: resident targets + gate table + work buffers (``slam_ctx``)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        _check(self._lib.slam_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self.n_targets = 0
        self.n_gates = 0

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.slam_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- residency -------------------------------------------------------
    def device_info(self) -> Tuple[str, int, int]:
        buf = C.create_string_buffer(256)
        cu, khz = C.c_int(0), C.c_int(0)
        _check(self._lib.slam_ctx_device_info(self._h, buf, 256, C.byref(cu), C.byref(khz)))
        return buf.value.decode(), cu.value, khz.value

    def set_targets(self, targets: np.ndarray) -> None:
        t = _mat_to_ri(targets)
        if t.ndim != 4:
            raise ValueError("targets must have shape [N, 4, 4]")
        _check(self._lib.slam_set_targets(self._h, _ptr(t), t.shape[0]))
        self.n_targets = t.shape[0]

    def c1c2c3(self, unitaries: np.ndarray, ndigits: int = 8) -> np.ndarray:
        """Weyl coordinates of ``unitaries[N, 4, 4]`` (weylchamber.c1c2c3, basis_abc.py:80-84) -> float64[N, 3]."""
        u = np.ascontiguousarray(unitaries, dtype=np.complex128)
        if u.ndim != 3 or u.shape[1:] != (4, 4):
            raise ValueError("expected an array of shape [N, 4, 4]")
        out = np.zeros((u.shape[0], 3), dtype=np.float64)
        _check(self._lib.slam_c1c2c3(self._h, _ptr(u.view(np.float64)), u.shape[0], int(ndigits), _ptr(out)))
        return out

    def targets_c1c2c3(self, first: int = 0, count: Optional[int] = None, ndigits: int = 8) -> np.ndarray:
        """Weyl coordinates of the resident targets [first, first + count) -> float64[count, 3]."""
        count = self.n_targets - first if count is None else count
        out = np.zeros((count, 3), dtype=np.float64)
        _check(self._lib.slam_targets_c1c2c3(self._h, int(first), int(count), int(ndigits), _ptr(out)))
        return out

    def predict_spans(self, gate_coords_seq, k_max: int, first: int = 0, count: Optional[int] = None, tol: float = 2e-8) -> np.ndarray:
        """Template size every resident target of [first, first + count) needs with the gate sequence whose Weyl coordinates are
        ``gate_coords_seq`` (0 local, 1..k_max, k_max + 1 out of reach): ``coverage.minimal_prefix`` evaluated on the device
        (slam_predict_spans) -- the half-spaces of the sequence's prefixes are computed here, the targets never leave the GPU."""
        from . import coverage

        g = np.asarray(gate_coords_seq, dtype=np.float64).reshape(-1, 3)
        if not 1 <= k_max <= min(len(g), MAX_SPAN_EVAL):
            raise ValueError(f"k_max must be 1..{min(len(g), MAX_SPAN_EVAL)}")
        count = self.n_targets - first if count is None else count
        point = np.ascontiguousarray(coverage.alcove_coordinates(g[:1])[0])
        bounds = np.full((k_max, len(coverage._PATTERNS)), -np.inf)
        for k in range(2, k_max + 1):
            bounds[k - 1] = coverage.region(g[:k])
        out = np.zeros(count, dtype=np.int32)
        _check(self._lib.slam_predict_spans(self._h, int(first), int(count), int(k_max), _ptr(point), _ptr(bounds), float(tol), _ptr(out)))
        return out

    def eval_c1c2c3(self, gate_seq: Sequence[int], x: np.ndarray, ndigits: int = 8) -> np.ndarray:
        """Weyl coordinates of CircuitTemplate.eval(x[m]) for ``x[M, n]`` (optimizer.py:85,103) -> float64[M, 3];
        the unitaries never leave the device."""
        k = len(gate_seq)
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != 6 * (k + 1):
            raise ValueError(f"x must have shape [M, {6 * (k + 1)}]")
        gs = np.ascontiguousarray(gate_seq, dtype=np.int32)
        out = np.zeros((x.shape[0], 3), dtype=np.float64)
        _check(self._lib.slam_eval_c1c2c3(self._h, k, _ptr(gs), _ptr(x), x.shape[0], int(ndigits), _ptr(out)))
        return out

    def sample_haar(self, seed: int, n_targets: int, first_index: int = 0) -> None:
        """Generate the resident batch on the device: T_i = Haar(seed, first_index + i)."""
        _check(self._lib.slam_sample_haar(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_index), int(n_targets)))
        self.n_targets = int(n_targets)

    def get_targets(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """Resident targets [first, first + count) as complex128[count, 4, 4]."""
        count = self.n_targets - first if count is None else count
        out = np.empty((count, 4, 4, 2), dtype=np.float64)
        _check(self._lib.slam_get_targets(self._h, int(first), int(count), _ptr(out)))
        return out.view(np.complex128).reshape(count, 4, 4)

    def set_gates(self, gates: np.ndarray) -> None:
        g = _mat_to_ri(gates)
        if g.ndim != 4:
            raise ValueError("gates must have shape [G, 4, 4]")
        # the same table as the last upload (one basis, many calls): nothing to do -- the upload is a synchronous copy (~25 us)
        last = getattr(self, "_gates_uploaded", None)
        if last is not None and last.shape == g.shape and np.array_equal(last, g):
            return
        _check(self._lib.slam_set_gates(self._h, _ptr(g), g.shape[0]))
        self.n_gates = g.shape[0]
        self._gates_uploaded = g.copy()

    # -- fused loss + gradient -------------------------------------------
    def eval_loss_grad(self, gate_seq: Sequence[int], x: np.ndarray, target_of: np.ndarray, want_grad=True):
        k = len(gate_seq)
        n = 6 * (k + 1)
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != n:
            raise ValueError(f"x must have shape [M, {n}]")
        M = x.shape[0]
        tof = np.ascontiguousarray(target_of, dtype=np.int32)
        if tof.shape != (M,):
            raise ValueError("target_of must have shape [M]")
        gs = np.ascontiguousarray(gate_seq, dtype=np.int32)
        loss = np.empty(M, dtype=np.float64)
        grad = np.empty((M, n), dtype=np.float64) if want_grad else None
        _check(self._lib.slam_eval_loss_grad(self._h, k, _ptr(gs), _ptr(x), _ptr(tof), M, _ptr(loss), _ptr(grad)))
        return loss, grad

    def eval_unitary(self, gate_seq: Sequence[int], x: np.ndarray, target_of: Optional[np.ndarray] = None):
        """W(x) for each row of x: complex128[M, 4, 4] (and BasicCost vs target_of if given)."""
        k = len(gate_seq)
        n = 6 * (k + 1)
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != n:
            raise ValueError(f"x must have shape [M, {n}]")
        M = x.shape[0]
        tof = np.zeros(M, np.int32) if target_of is None else np.ascontiguousarray(target_of, dtype=np.int32)
        gs = np.ascontiguousarray(gate_seq, dtype=np.int32)
        w = np.empty((M, 4, 4, 2), dtype=np.float64)
        loss = np.empty(M, dtype=np.float64)
        _check(self._lib.slam_eval_unitary(self._h, k, _ptr(gs), _ptr(x), _ptr(tof), M, _ptr(w), _ptr(loss)))
        return w.view(np.complex128).reshape(M, 4, 4), loss

    # -- one span stage ----------------------------------------------------
    def minimize_stage(
        self,
        gate_seq: Sequence[int],
        params: OptParams,
        active: Optional[np.ndarray] = None,
        x0: Optional[np.ndarray] = None,
        want_items: bool = True,
    ) -> dict:
        k = len(gate_seq)
        n = 6 * (k + 1)
        gs = np.ascontiguousarray(gate_seq, dtype=np.int32)
        if active is not None:
            active = np.ascontiguousarray(active, dtype=np.int32)
            na = active.shape[0]
        else:
            na = self.n_targets
        R = int(params.restarts)
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            if x0.shape != (na, R, n):
                raise ValueError(f"x0 must have shape [{na}, {R}, {n}]")
        out = {
            "best_loss": np.empty(na, dtype=np.float64),
            "best_x": np.empty((na, n), dtype=np.float64),
            "best_restart": np.empty(na, dtype=np.int32),
        }
        if want_items:
            out["item_loss"] = np.empty((na, R), dtype=np.float64)
            out["item_iters"] = np.empty((na, R), dtype=np.int32)
            out["item_status"] = np.empty((na, R), dtype=np.int32)
            out["item_evals"] = np.empty((na, R), dtype=np.int32)
        _check(
            self._lib.slam_minimize_stage(
                self._h, k, _ptr(gs), _ptr(active), na, _ptr(x0), C.byref(params),
                _ptr(out["best_loss"]), _ptr(out["best_x"]), _ptr(out["best_restart"]),
                _ptr(out.get("item_loss")), _ptr(out.get("item_iters")), _ptr(out.get("item_status")),
                _ptr(out.get("item_evals")),
            )
        )
        return out

    # -- the whole span loop -------------------------------------------------
    @staticmethod
    def _flat_gate_seqs(gate_seqs: Sequence[Sequence[int]], k_min: int, k_max: int) -> np.ndarray:
        if len(gate_seqs) != k_max - k_min + 1:
            raise ValueError("need one gate sequence per span")
        flat = []
        for k, gs in zip(range(k_min, k_max + 1), gate_seqs):
            if len(gs) != k:
                raise ValueError(f"gate sequence for span {k} has length {len(gs)}")
            flat.extend(int(g) for g in gs)
        return np.asarray(flat, dtype=np.int32)

    def decompose(self, k_min, k_max, gate_seqs, params: OptParams, success_threshold: float, fetch=True):
        flat = self._flat_gate_seqs(gate_seqs, k_min, k_max)
        _check(self._lib.slam_decompose_resident(self._h, k_min, k_max, _ptr(flat), C.byref(params), float(success_threshold)))
        if fetch:
            return self.fetch_results(k_max)
        return None

    def fetch_results(self, k_max: int):
        nmax = 6 * (k_max + 1)
        N = self.n_targets
        best_loss = np.empty(N, dtype=np.float64)
        best_x = np.zeros((N, nmax), dtype=np.float64)
        best_cycles = np.empty(N, dtype=np.int32)
        _check(self._lib.slam_fetch_results(self._h, k_max, _ptr(best_loss), _ptr(best_x), _ptr(best_cycles)))
        return best_loss, best_x, best_cycles

    def decompose_range(self, first, count, k_min, k_max, gate_seqs, params: OptParams, success_threshold: float, fetch=True, pinned=False):
        flat = self._flat_gate_seqs(gate_seqs, k_min, k_max)
        if not fetch or not hasattr(self._lib, "slam_decompose_range_fetch"):
            _check(self._lib.slam_decompose_range(self._h, int(first), int(count), k_min, k_max, _ptr(flat), C.byref(params), float(success_threshold)))
            return self.fetch_results_range(k_max, first, count, pinned=pinned) if fetch else None
        nmax = 6 * (k_max + 1)
        new = result_pool.empty if pinned else pageable_pool.empty
        best_loss = new(count, np.float64)
        best_x = new((count, nmax), np.float64)  # (every row is written: the resident rows are zero-padded)
        best_cycles = new(count, np.int32)
        _check(self._lib.slam_decompose_range_fetch(self._h, int(first), int(count), k_min, k_max, _ptr(flat), C.byref(params),
                                                     float(success_threshold), _ptr(best_loss), _ptr(best_x), _ptr(best_cycles)))
        return best_loss, best_x, best_cycles

    def decompose_list(self, targets, k_min, k_max, gate_seqs, params: OptParams, success_threshold: float, k_layout: int = 0):
        """Span loop for an explicit list of resident-target indices (slam_decompose_list); results stay in the
        per-target resident arrays (row width 6 (k_layout + 1)): fetch with ``fetch_results_range(k_layout, ...)``."""
        idx = np.ascontiguousarray(targets, dtype=np.int32)
        flat = self._flat_gate_seqs(gate_seqs, k_min, k_max)
        _check(self._lib.slam_decompose_list(self._h, _ptr(idx), idx.shape[0], k_min, k_max, int(k_layout), _ptr(flat),
                                              C.byref(params), float(success_threshold)))

    def decompose_predicted(self, gate_coords_seq, k_max: int, gate_seqs, params: OptParams, success_threshold: float, first: int = 0,
                            count: Optional[int] = None, carry: bool = False, tol: float = 2e-8):
        """``use_polytopes`` mode for the resident targets [first, first + count) in one chain of kernels (slam_decompose_predicted): the
        coverage lookup of ``predict_spans``, per-size target lists and the span loop, all on the device.  ``gate_seqs``: the sequences
        of spans 1..k_max.  Results stay resident (``fetch_results_range(k_max, ...)``).  Returns (n_local, n_unreachable)."""
        from . import coverage

        g = np.asarray(gate_coords_seq, dtype=np.float64).reshape(-1, 3)
        if not 1 <= k_max <= min(len(g), MAX_SPAN_MINIMIZE):
            raise ValueError(f"k_max must be 1..{min(len(g), MAX_SPAN_MINIMIZE)}")
        count = self.n_targets - first if count is None else count
        point = np.ascontiguousarray(coverage.alcove_coordinates(g[:1])[0])
        bounds = np.full((k_max, len(coverage._PATTERNS)), -np.inf)
        for k in range(2, k_max + 1):
            bounds[k - 1] = coverage.region(g[:k])
        flat = self._flat_gate_seqs(gate_seqs, 1, k_max)
        n_loc, n_unr = C.c_int64(0), C.c_int64(0)
        _check(self._lib.slam_decompose_predicted(self._h, int(first), int(count), int(k_max), _ptr(point), _ptr(bounds), float(tol), int(bool(carry)),
                                                  _ptr(flat), C.byref(params), float(success_threshold), C.byref(n_loc), C.byref(n_unr)))
        return int(n_loc.value), int(n_unr.value)

    def fetch_results_range(self, k_max: int, first: int, count: int, pinned: bool = False):
        nmax = 6 * (k_max + 1)
        new = result_pool.empty if pinned else pageable_pool.empty
        best_loss = new(count, np.float64)
        best_x = new((count, nmax), np.float64)
        best_cycles = new(count, np.int32)
        _check(self._lib.slam_fetch_results_range(self._h, k_max, int(first), int(count), _ptr(best_loss), _ptr(best_x), _ptr(best_cycles)))
        return best_loss, best_x, best_cycles

    def fetch_span_losses(self, first: int, count: int) -> np.ndarray:
        """Running best loss after every span the last span loop ran: float64[count, MAX_SPAN_EVAL] (NaN = span not run)."""
        out = np.empty((count, MAX_SPAN_EVAL), dtype=np.float64)
        _check(self._lib.slam_fetch_span_losses(self._h, int(first), int(count), _ptr(out)))
        return out

    def minimize_stage_trace(self, gate_seq: Sequence[int], params: OptParams, exit_loss: float, trace_cap: int,
                             active: Optional[np.ndarray] = None, x0: Optional[np.ndarray] = None) -> dict:
        """``minimize_stage`` plus the per-iteration trajectories of every restart (use_callback, optimizer.py:217-224):
        ``trace_loss[na, R, cap]`` and ``trace_x[na, R, cap, n]`` (NaN beyond ``item_iters``)."""
        k = len(gate_seq)
        n = 6 * (k + 1)
        gs = np.ascontiguousarray(gate_seq, dtype=np.int32)
        if active is not None:
            active = np.ascontiguousarray(active, dtype=np.int32)
            na = active.shape[0]
        else:
            na = self.n_targets
        R = int(params.restarts)
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            if x0.shape != (na, R, n):
                raise ValueError(f"x0 must have shape [{na}, {R}, {n}]")
        cap = int(trace_cap)
        out = {
            "best_loss": np.empty(na, dtype=np.float64),
            "best_x": np.empty((na, n), dtype=np.float64),
            "best_restart": np.empty(na, dtype=np.int32),
            "item_loss": np.empty((na, R), dtype=np.float64),
            "item_iters": np.empty((na, R), dtype=np.int32),
            "item_status": np.empty((na, R), dtype=np.int32),
            "trace_loss": np.empty((na, R, cap), dtype=np.float64),
            "trace_x": np.empty((na, R, cap, n), dtype=np.float64),
        }
        _check(
            self._lib.slam_minimize_stage_trace(
                self._h, k, _ptr(gs), _ptr(active), na, _ptr(x0), C.byref(params), float(exit_loss), cap,
                _ptr(out["best_loss"]), _ptr(out["best_x"]), _ptr(out["best_restart"]), _ptr(out["item_loss"]),
                _ptr(out["item_iters"]), _ptr(out["item_status"]), _ptr(out["trace_loss"]), _ptr(out["trace_x"]),
            )
        )
        return out

    # -- templates with parametrised 2Q gates (CircuitTemplateV2) ------------------------
    def v2_set_gates(self, gates: Sequence["V2Gate"]) -> None:
        arr = (V2Gate * len(gates))(*gates)
        _check(self._lib.slam_v2_set_gates(self._h, arr, len(gates)))
        self.v2_qn = int(gates[0].n_params)

    def v2_set_constraint(self, k: int, weights: Optional[np.ndarray], cost_max: float = 0.0) -> None:
        """sum_i weights[i] x_i <= cost_max for every later stage of span k (device parameter order); None removes it.
        ``v2_set_gates`` removes the constraints of every span."""
        if weights is None:
            _check(self._lib.slam_v2_set_constraint(self._h, int(k), None, 0, 0.0))
            return
        w = np.ascontiguousarray(weights, dtype=np.float64)
        _check(self._lib.slam_v2_set_constraint(self._h, int(k), w.ctypes.data, int(w.size), float(cost_max)))

    def v2_eval(self, gate_seq: Sequence[int], x: np.ndarray, target_of: Optional[np.ndarray] = None, want_grad=True, want_unitary=False):
        """Loss, gradient w.r.t. all n = 6 (k + 1) + QN k parameters and (optionally) W(x) for ``x[M, n]``."""
        k = len(gate_seq)
        n = 6 * (k + 1) + self.v2_qn * k
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim != 2 or x.shape[1] != n:
            raise ValueError(f"x must have shape [M, {n}]")
        M = x.shape[0]
        tof = np.zeros(M, np.int32) if target_of is None else np.ascontiguousarray(target_of, dtype=np.int32)
        gs = np.ascontiguousarray(gate_seq, dtype=np.int32)
        loss = np.empty(M, dtype=np.float64)
        grad = np.empty((M, n), dtype=np.float64) if want_grad else None
        w = np.empty((M, 4, 4, 2), dtype=np.float64) if want_unitary else None
        _check(self._lib.slam_v2_eval_loss_grad(self._h, k, _ptr(gs), _ptr(x), _ptr(tof), M, _ptr(loss), _ptr(grad), _ptr(w)))
        return loss, grad, (w.view(np.complex128).reshape(M, 4, 4) if want_unitary else None)

    def v2_minimize_stage(self, gate_seq: Sequence[int], params: OptParams, exit_loss: float, init_lo, init_hi, bound_lo=None,
                          bound_hi=None, active: Optional[np.ndarray] = None, x0: Optional[np.ndarray] = None, want_items: bool = True) -> dict:
        k = len(gate_seq)
        n = 6 * (k + 1) + self.v2_qn * k
        gs = np.ascontiguousarray(gate_seq, dtype=np.int32)
        if active is not None:
            active = np.ascontiguousarray(active, dtype=np.int32)
            na = active.shape[0]
        else:
            na = self.n_targets
        R = int(params.restarts)
        vecs = []
        for v in (init_lo, init_hi, bound_lo, bound_hi):
            if v is not None:
                v = np.ascontiguousarray(v, dtype=np.float64)
                if v.shape != (n,):
                    raise ValueError(f"per-parameter arrays must have shape [{n}]")
            vecs.append(v)
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, dtype=np.float64)
            if x0.shape != (na, R, n):
                raise ValueError(f"x0 must have shape [{na}, {R}, {n}]")
        out = {
            "best_loss": np.empty(na, dtype=np.float64),
            "best_x": np.empty((na, n), dtype=np.float64),
            "best_restart": np.empty(na, dtype=np.int32),
            "item_loss": np.empty((na, R), dtype=np.float64) if want_items else None,
            "item_iters": np.empty((na, R), dtype=np.int32) if want_items else None,
            "item_status": np.empty((na, R), dtype=np.int32) if want_items else None,
            "item_evals": np.empty((na, R), dtype=np.int32) if want_items else None,
        }
        _check(
            self._lib.slam_v2_minimize_stage(
                self._h, k, _ptr(gs), _ptr(active), na, _ptr(x0), _ptr(vecs[0]), _ptr(vecs[1]), _ptr(vecs[2]), _ptr(vecs[3]),
                C.byref(params), float(exit_loss), _ptr(out["best_loss"]), _ptr(out["best_x"]), _ptr(out["best_restart"]),
                _ptr(out["item_loss"]), _ptr(out["item_iters"]), _ptr(out["item_status"]), _ptr(out["item_evals"]),
            )
        )
        return out

    def v2_decompose_range(self, first: int, count: int, k_min: int, k_max: int, gate_seqs, layouts, params: OptParams, threshold: float):
        """The span loop of a V2 template on the device (``slam_v2_decompose_range``).  ``layouts[i]`` = (init_lo, init_hi, bound_lo,
        bound_hi) of span ``k_min + i`` in device order.  Returns (best_loss [count], best_x [count, n_kmax], best_cycles [count]); a
        row holds the n_k parameters of its target's span k in front, zeros behind."""
        qn = self.v2_qn
        nmax = 6 * (k_max + 1) + qn * k_max
        gs = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.int32) for g in gate_seqs]))
        cat = [np.ascontiguousarray(np.concatenate([np.asarray(l[j], dtype=np.float64) for l in layouts])) for j in range(4)]
        n_tot = sum(6 * (k + 1) + qn * k for k in range(k_min, k_max + 1))
        if len(gs) != sum(range(k_min, k_max + 1)) or any(len(v) != n_tot for v in cat):
            raise ValueError("gate_seqs / layouts do not match the span range")
        best_loss = pageable_pool.empty(count, np.float64)
        best_x = pageable_pool.empty((count, nmax), np.float64)
        best_cycles = pageable_pool.empty(count, np.int32)
        _check(self._lib.slam_v2_decompose_range(self._h, int(first), int(count), int(k_min), int(k_max), _ptr(gs), _ptr(cat[0]), _ptr(cat[1]),
                                                 _ptr(cat[2]), _ptr(cat[3]), C.byref(params), float(threshold), _ptr(best_loss), _ptr(best_x),
                                                 _ptr(best_cycles)))
        return best_loss, best_x, best_cycles

    def v2_minimize_stage_trace(self, gate_seq: Sequence[int], params: OptParams, exit_loss: float, trace_cap: int, init_lo, init_hi,
                                bound_lo=None, bound_hi=None, active: Optional[np.ndarray] = None) -> dict:
        """``v2_minimize_stage`` plus the loss / parameters after every accepted iteration of every restart
        (``trace_loss`` [na, R, cap], ``trace_x`` [na, R, cap, n]; NaN beyond an item's iterations)."""
        k = len(gate_seq)
        n = 6 * (k + 1) + self.v2_qn * k
        gs = np.ascontiguousarray(gate_seq, dtype=np.int32)
        if active is not None:
            active = np.ascontiguousarray(active, dtype=np.int32)
            na = active.shape[0]
        else:
            na = self.n_targets
        R, cap = int(params.restarts), int(trace_cap)
        vecs = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in (init_lo, init_hi, bound_lo, bound_hi)]
        for v in vecs:
            if v is not None and v.shape != (n,):
                raise ValueError(f"per-parameter arrays must have shape [{n}]")
        out = {
            "best_loss": np.empty(na, dtype=np.float64), "best_x": np.empty((na, n), dtype=np.float64),
            "best_restart": np.empty(na, dtype=np.int32), "item_loss": np.empty((na, R), dtype=np.float64),
            "item_iters": np.empty((na, R), dtype=np.int32), "item_status": np.empty((na, R), dtype=np.int32),
            "trace_loss": np.empty((na, R, cap), dtype=np.float64), "trace_x": np.empty((na, R, cap, n), dtype=np.float64),
        }
        _check(self._lib.slam_v2_minimize_stage_trace(
            self._h, k, _ptr(gs), _ptr(active), na, None, _ptr(vecs[0]), _ptr(vecs[1]), _ptr(vecs[2]), _ptr(vecs[3]), C.byref(params),
            float(exit_loss), cap, _ptr(out["best_loss"]), _ptr(out["best_x"]), _ptr(out["best_restart"]), _ptr(out["item_loss"]),
            _ptr(out["item_iters"]), _ptr(out["item_status"]), _ptr(out["trace_loss"]), _ptr(out["trace_x"])))
        return out

    def set_cost(self, kind: int) -> None:
        """0 = BasicCost (default), 1 = SquareCost."""
        _check(self._lib.slam_set_cost(self._h, int(kind)))

    def synchronize(self) -> None:
        _check(self._lib.slam_synchronize(self._h))

    def stats(self) -> dict:
        s = Stats()
        _check(self._lib.slam_get_stats(self._h, C.byref(s)))
        return {
            "kernel_ms": s.kernel_ms,
            "kernel_launches": s.kernel_launches,
            "evals": list(s.evals),
            "items": list(s.items),
            "total_ms": s.total_ms,
            "kernel_ms_span": list(s.kernel_ms_span),
            "wave_rounds": list(s.wave_rounds),
            "evals_accepted": list(s.evals_accepted),
            "evals_preempted": list(s.evals_preempted),
        }

    def reset_stats(self) -> None:
        _check(self._lib.slam_reset_stats(self._h))

    def best_loss_device_ptr(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_int64(0)
        _check(self._lib.slam_best_loss_device_ptr(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)


def decompose_multi(ctxs: Sequence["Context"], first: int, count: int, k_min: int, k_max: int, gate_seqs, params: OptParams,
                    success_threshold: float) -> None:
    """The span loops of several contexts (same device, same target window, each its own gate table) as ONE chain of kernels
    (``slam_decompose_multi``): per span one multi-queue optimizer launch and one bookkeeping launch for all of them.  Results
    stay resident in each context: ``ctx.fetch_results_range(k_max, first, count)``."""
    lib = load_library()
    flat = Context._flat_gate_seqs(gate_seqs, k_min, k_max)
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(lib.slam_decompose_multi(arr, len(ctxs), int(first), int(count), k_min, k_max, _ptr(flat), C.byref(params), float(success_threshold)))


class Comm:
    """RCCL communicator of one rank (``slam_comm``): one process per GPU, xGMI underneath.  Only the job's final
    best-loss min-all-reduce and a few scalars go through it."""

    def __init__(self, device: int, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        self._lib = load_library()
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _check(self._lib.slam_comm_init(int(device), int(rank), int(world), buf, C.byref(self._h)))
        self.rank, self.world, self.device = int(rank), int(world), int(device)

    @staticmethod
    def unique_id() -> bytes:
        """``ncclGetUniqueId`` (rank 0 calls this and hands the bytes to the other ranks)."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(load_library().slam_comm_get_unique_id(buf))
        return buf.raw

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.slam_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _allreduce(self, a: np.ndarray, op: int) -> None:
        if a.dtype != np.float64 or not a.flags.c_contiguous:
            raise ValueError("all-reduce buffers must be C-contiguous float64")
        _check(self._lib.slam_comm_allreduce_f64(self._h, _ptr(a), a.size, op))

    def allreduce_min(self, a: np.ndarray) -> None:
        self._allreduce(a, OP_MIN)

    def allreduce_max(self, a: np.ndarray) -> None:
        self._allreduce(a, OP_MAX)

    def allreduce_sum(self, a: np.ndarray) -> None:
        self._allreduce(a, OP_SUM)

    def barrier(self) -> None:
        _check(self._lib.slam_comm_barrier(self._h))

    def rccl_rank_world(self):
        """(rank, world size) as RCCL itself reports them for this communicator (``ncclCommUserRank`` / ``ncclCommCount``)."""
        r, w = C.c_int(-1), C.c_int(-1)
        _check(self._lib.slam_comm_rank(self._h, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def merge_begin(self, n_global: int) -> None:
        _check(self._lib.slam_comm_merge_begin(self._h, int(n_global)))
        self._merge_n = int(n_global)

    def merge_add(self, ctx: "Context", first_local: int, count: int, first_global: int) -> None:
        """Min-merge the context's resident best_loss window into the job-wide vector, device to device."""
        _check(self._lib.slam_comm_merge_add(self._h, ctx._h, int(first_local), int(count), int(first_global)))

    def merge_add_host(self, loss: np.ndarray, first_global: int) -> None:
        loss = np.ascontiguousarray(loss, dtype=np.float64)
        _check(self._lib.slam_comm_merge_add_host(self._h, _ptr(loss), loss.size, int(first_global)))

    def allreduce_min_merged(self, threshold: float, want_merged: bool = False):
        """The job's one collective.  Returns (number of entries < threshold, merged vector or None).  The host copy is
        sized from the ``n_global`` given to ``merge_begin`` (the library checks the capacity again)."""
        nb = C.c_int64(0)
        n = getattr(self, "_merge_n", 0)
        if n <= 0:
            raise RuntimeError("allreduce_min_merged: call merge_begin first")
        merged = np.empty(n, dtype=np.float64) if want_merged else None
        _check(self._lib.slam_allreduce_min(self._h, float(threshold), C.byref(nb), _ptr(merged), n if want_merged else 0))
        return int(nb.value), merged
