"""CPU: the C-ABI shared library loads and exports every symbol include/slam_hip.h declares;
argument validation that needs no GPU."""
import ctypes
import os
import re

import pytest

from slam_decomposition_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# a machine without ROCm (a hosted CI runner) can neither build nor load the library; everywhere else a missing
# libslamhip.so is a failure, not a skip
pytestmark = pytest.mark.skipif(
    not os.path.exists(_ffi.LIB_PATH) and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no ROCm toolchain on this machine"
)


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "slam_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported_and_bound():
    lib = _ffi.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in slam_hip.h but not exported"
    assert sorted(_ffi.EXPORTED_SYMBOLS) == declared


def test_version_and_struct_layout():
    lib = _ffi.load_library()
    assert b"gfx950" in lib.slam_version()
    # slam_opt_params: i32 i32 f64 f64 u64 u32 u32 f64 f64 i64
    assert ctypes.sizeof(_ffi.OptParams) == 64
    assert _ffi.OptParams.target_base.offset == 56
    assert _ffi.OptParams.gtol_far.offset == 40
    p = _ffi.OptParams(restarts=7, seed=2**63 + 5)
    assert p.restarts == 7 and p.seed == 2**63 + 5 and p.gtol_far == 1e-5 and p.far_loss == 1e-6
    w = _ffi.MAX_SPAN_EVAL + 1  # per-span arrays: index k = 0 .. SLAM_MAX_SPAN_EVAL (ABI 6: 16)
    assert _ffi.MAX_SPAN_EVAL == 16 and _ffi.MAX_SPAN_MINIMIZE == 16 and _ffi.MAX_SPAN_QUAD == 5
    assert ctypes.sizeof(_ffi.Stats) == 8 + 8 + w * 8 + w * 8 + 8 + w * 8 + w * 8 + w * 8 + w * 8
    assert lib.slam_abi_version() == _ffi.ABI_VERSION == 7


def test_no_gpu_fails_loudly():
    """Without a usable GPU the product path raises (no CPU fallback)."""
    n = ctypes.c_int(-1)
    rc = _ffi.load_library().slam_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_ffi.SlamHipError):
        _ffi.Context(0)
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.optimizer import TemplateOptimizer

    opt = TemplateOptimizer(CircuitTemplate(maximum_span_guess=2), BasicCost(), override_fail=True)
    import numpy as np

    with pytest.raises(_ffi.SlamHipError):
        opt.approximate_target_U(np.eye(4))


def test_product_package_does_not_import_the_oracle():
    import subprocess
    import sys

    code = (
        "import sys; import slam_decomposition_amd.optimizer, slam_decomposition_amd.basis, "
        "slam_decomposition_amd.sampler, slam_decomposition_amd.parallel; "
        "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'; "
        "assert 'torch' not in sys.modules, 'torch imported'"
    )
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_result_pool_falls_back_to_pageable_arrays_without_a_gpu_and_recycles_blocks():
    """``_ffi.PinnedPool``: small arrays are ordinary NumPy arrays; big ones come from slam_host_alloc, and where that fails (no GPU
    here) the pool hands out pageable arrays instead of failing (slower, not wrong).  Recycling is exercised with a stand-in allocator."""
    import ctypes as C

    import numpy as np

    pool = _ffi.ResultPool(pinned=True)
    a = pool.empty((10, 3), np.float64)
    assert a.shape == (10, 3) and a.dtype == np.float64 and a.base is None
    b = pool.empty((1 << 16, 24), np.float64)  # 12.6 MB: asks the library, which has no device here
    assert b.shape == (1 << 16, 24) and b.flags["C_CONTIGUOUS"] and b.flags["WRITEABLE"]
    b[:] = 1.0

    class FakeLib:
        def __init__(self):
            self.bufs, self.freed = {}, []

        def slam_host_alloc(self, n, out):
            buf = C.create_string_buffer(n)
            self.bufs[C.addressof(buf)] = buf
            C.cast(out, C.POINTER(C.c_void_p))[0] = C.addressof(buf)
            return 0

        def slam_host_free(self, p):
            self.freed.append(p.value)
            return 0

    fake = FakeLib()
    orig = _ffi.load_library
    _ffi.load_library = lambda: fake
    try:
        pool = _ffi.ResultPool(pinned=True)
        x = pool.empty((1 << 15, 24), np.float64)
        x[:] = 3.0
        view = x[5:7]
        addr = x.ctypes.data
        del x
        assert pool.allocated == 1 and pool._idle == 0 and float(view[0, 0]) == 3.0  # a view keeps the block out of the pool
        del view
        assert pool._idle > 0
        y = pool.empty((1 << 15, 24), np.float64)
        assert y.ctypes.data == addr and pool.allocated == 1  # the same block again
        z = pool.empty(1 << 20, np.int32)
        assert pool.allocated == 2 and z.dtype == np.int32 and z.shape == (1 << 20,)
        pool.MAX_IDLE_BYTES = 0
        del y, z
        assert len(fake.freed) == 2  # beyond the idle cap blocks go back to the driver
    finally:
        _ffi.load_library = orig
    # the pageable pool: ordinary NumPy buffers, recycled the same way (nothing goes back to the C allocator between calls)
    pool = _ffi.ResultPool(pinned=False)
    p1 = pool.empty((1 << 15, 24), np.float64)
    p1[:] = 2.0
    addr = p1.ctypes.data
    tail = p1[-1]
    del p1
    assert pool._idle == 0 and float(tail[3]) == 2.0
    del tail
    p2 = pool.empty((1 << 15, 24), np.float64)
    assert p2.ctypes.data == addr and pool.allocated == 1
