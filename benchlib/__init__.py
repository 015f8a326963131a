"""The pieces of bench.py (repo root): workloads, CPU baseline + parity sample, launcher, secondary probes, the timed region and CLI."""
