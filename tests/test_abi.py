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
    assert lib.slam_abi_version() == _ffi.ABI_VERSION == 6


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
