python bench.py --workload batching_sorted --sorted-items 4096 --steps 50 --warmup 10 --blocks 4 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic
