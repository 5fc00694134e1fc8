python bench.py --workload tree --tree-cull --tree-cull-launches 2 --steps 50 --warmup 10 --blocks 4 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic
