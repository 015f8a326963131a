"""CPU: the exchange-area layout of the optimizer kernel (csrc/slam_device.hpp, Cfg<K, true>) against the LDS bank model of
MI355X_MICROARCH.md (tools/lds_bank_model.py): the shipped quad strides and per-quad trig offsets are free of modelled bank
conflicts (measured on the GPU: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 33 % -> 8-10 %), and the constants in the header
are the ones the model was run with."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(ROOT, "tools", "lds_bank_model.py"))
lds = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lds)


def header_layout(K):
    """Cfg<K, true> recomputed from the formulas in slam_device.hpp (the regexes fail loudly if they change)."""
    src = open(os.path.join(ROOT, "slam_decomposition_amd", "csrc", "slam_device.hpp")).read()
    assert re.search(r"TOFF = PSQ \? 6 : 0;", src)
    assert re.search(r"PSP = \(N \+ 3\) / 4 \* 4 \+ 1;", src)
    assert re.search(r"PS0 = 12 \* K \+ TOFF;", src)
    assert re.search(r"XSTRIDE = PSQ \? \(XNEED - 4 \+ 15\) / 16 \* 16 \+ 4", src)
    assert re.search(r"\(threadIdx\.x & 4\) \+ \(\(threadIdx\.x >> 2\) & 2\)", src)  # offsets (0, 4, 2, 6) by quad mod 4
    N = 6 * (K + 1)
    NA = (N + 3) // 4
    fstride = (20 * NA - 16 - 4 + 31) // 32 * 32 + 4
    xneed = max(12 * K + 6 + 3 * ((N + 3) // 4 * 4 + 1) + N, (16 * fstride + 31) // 32)
    return (xneed - 4 + 15) // 16 * 16 + 4, xneed


def test_shipped_exchange_area_layout_has_no_modelled_bank_conflicts():
    for K, want_stride in ((1, 84), (2, 116), (3, 148)):
        stride, xneed = header_layout(K)
        assert stride == want_stride
        conflicts, need = lds.model(K, stride, (0, 4, 2, 6))
        assert need <= stride and need == xneed
        assert sum(conflicts.values()) == 0, (K, conflicts)
        # the layouts it replaced, for the record: a uniform stride conflicts on the trig stores, an offset for the odd
        # quads alone on the broadcast reads
        assert lds.model(K, stride, (0, 0, 0, 0))[0]["trig_w"] > 0
        assert lds.model(K, stride, (0, 4, 0, 4))[0]["trig_r"] > 0


def test_eight_wavefronts_per_cu_fit_at_span_two():
    stride, _ = header_layout(2)
    lds_bytes = (16 * stride + (2 - 1) * 4 * 65 * 2 + 128 + 4) * 8  # exchange area + stored vector + sincos table + thresholds
    assert 8 * lds_bytes <= 160 * 1024
