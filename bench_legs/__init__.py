"""The legs of bench.py, one module per workload family (round 6: bench.py itself is the driver -- arguments, ranks, the one JSON line)."""
