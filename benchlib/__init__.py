"""bench.py's parts: one module per workload, the timing loop, the CPU baselines, the end-to-end block, the result line."""
