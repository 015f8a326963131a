"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SLAM template-optimizer hot path.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the timed CPU baseline.  The product
package ``slam_decomposition_amd`` never imports this package.

Parity pinning: the reference has no tests (``src/tests/main_test.py:4-6`` is a
placeholder) and cannot be imported here (qiskit / weylchamber / qutip /
monodromy are absent -- ordinary ``ModuleNotFoundError``; no permission was
denied).  The oracle is therefore a NumPy/SciPy restatement pinned by the
reference's recorded notebook outputs; see
``tests/test_oracle_kat.py``.

Files: ``slam_oracle.py`` (template, loss, gradient, Haar sampler, ``c1c2c3``, the reference's restart loop on SciPy),
``v2_oracle.py`` (templates with parametrised gates), ``bfgs_port.py`` (the kernel's quasi-Newton loop in NumPy),
``pqn_port.py`` (its projected variant for box bounds and the multiplier method for a cost constraint).
"""
