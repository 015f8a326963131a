"""slam_decomposition_amd -- MI355X-native batched template optimizer (SLAM hot path).

Python surface mirrors the reference (``TemplateOptimizer``, ``CircuitTemplate``,
``BasicCost``, ``HaarSample``); all numerics of the optimizer inner loop run in
hand-written HIP kernels behind the C ABI of ``include/slam_hip.h``.
"""
__version__ = "0.1.0"
