"""CPU: AddressSanitizer + UBSan run of libslamhip's host side (tools/asan_host.sh, `make asan`): every C-ABI entry
point through its argument-validation / no-device paths, including concurrent failing calls (thread-local message)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_side_under_address_sanitizer():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "asan host driver: ok" in r.stdout
