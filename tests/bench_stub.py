"""Stand-in for ``_ffi.Context`` / ``_ffi.device_count`` used ONLY by tests/test_bench_cpu.py through bench.py's
SLAM_BENCH_TEST_STUB hook: lets the multi-rank plumbing of bench.py (launcher, ranks, communicator, merge arithmetic,
JSON line) run on a box without a GPU.  It computes nothing; every "decomposition" is a fixed fake result derived from
the target's global index, so that the merged vector can be checked."""
import numpy as np


class StubContext:
    def __init__(self, device=0):
        self.device = device
        self.n_targets = 0
        self._seed0 = 0
        self._st = None
        self.reset_stats()

    def device_info(self):
        return "stub-device", 1, 0

    def set_gates(self, table):
        pass

    def set_targets(self, t):
        self.n_targets = len(t)

    def sample_haar(self, seed0, n, first_index=0):
        self._seed0, self.n_targets = int(seed0), int(n)

    def predict_spans(self, gate_coords_seq, k_max, first=0, count=None, tol=2e-8):
        n = self.n_targets - first if count is None else count
        return np.full(n, k_max, dtype=np.int32)  # (every stand-in target "needs" the last span)

    def reset_stats(self):
        self._st = {"kernel_ms": 0.0, "kernel_launches": 0, "evals": [0] * 6, "items": [0] * 6, "evals_accepted": [0] * 6,
                    "evals_preempted": [0] * 6, "kernel_ms_span": [0.0] * 6, "wave_rounds": [0] * 6}

    def stats(self):
        return {k: (list(v) if isinstance(v, list) else v) for k, v in self._st.items()}

    def decompose_range(self, first, count, k_min, k_max, gate_seqs, prm, threshold, fetch=True):
        idx = self._seed0 + first + np.arange(count)
        # every 7th target (by GLOBAL index) "fails": the solved count of the merged vector is then checkable
        loss = np.where(idx % 7 == 0, 1e-3, 1e-12 * (1 + idx % 5)).astype(np.float64)
        cyc = np.where(idx % 7 == 0, 3, 2).astype(np.int32)
        for k in (1, 2, 3):
            self._st["evals"][k] += 40 * count * prm.restarts
            self._st["evals_accepted"][k] += 38 * count * prm.restarts
            self._st["items"][k] += count * prm.restarts
            self._st["wave_rounds"][k] += 3 * count * prm.restarts
            self._st["kernel_ms_span"][k] += 0.01
        self._st["kernel_ms"] += 0.03
        self._st["kernel_launches"] += 3
        return loss, np.zeros((count, 24)), cyc

    def best_loss_window(self, first, count):
        """What the resident best-loss array holds for targets [first, first + count) after their step ran."""
        idx = self._seed0 + first + np.arange(count)
        return np.where(idx % 7 == 0, 1e-3, 1e-12 * (1 + idx % 5)).astype(np.float64)

    def synchronize(self):
        pass

    def close(self):
        pass


class StubRaw:
    """Stand-in for ``_ffi.Comm``'s device-to-device merge (slam_comm_merge_begin / _add / slam_allreduce_min) on top of a host
    communicator: the same calls with the same slice arguments, the vector kept on the host."""

    def __init__(self, comm):
        self.comm = comm
        self.buf = None
        self.calls = 0

    def merge_begin(self, n_global):
        self.buf = np.full(int(n_global), np.inf)

    def merge_add(self, ctx, first_local, count, first_global):
        assert np.all(np.isinf(self.buf[first_global : first_global + count])), "two windows on the same slice of the job vector"
        self.buf[first_global : first_global + count] = ctx.best_loss_window(first_local, count)
        self.calls += 1

    def allreduce_min_merged(self, threshold, want_merged=False):
        self.comm.allreduce_min(self.buf)
        assert not np.any(np.isinf(self.buf)), "a slice of the job vector was contributed by no rank"
        return int((self.buf < threshold).sum()), (self.buf if want_merged else None)


def install():
    import os

    from slam_decomposition_amd import _ffi, parallel

    _ffi.Context = StubContext
    _ffi.device_count = lambda: 1
    if os.environ.get("SLAM_BENCH_STUB_RAW"):
        # rehearse bench.py's `resident_merge` branch (the one an RCCL run takes) over the file communicator
        init = parallel.FileComm.__init__

        def init_with_raw(self, *a, **kw):
            init(self, *a, **kw)
            self.raw = StubRaw(self)

        parallel.FileComm.__init__ = init_with_raw
