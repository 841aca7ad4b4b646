"""Benchmark legs behind bench.py (kept out of the driver script: roofline, families, CPU baselines, RMSE, other shapes)."""
