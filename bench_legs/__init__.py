"""The legs of bench.py (repo root), one module per workload family."""
