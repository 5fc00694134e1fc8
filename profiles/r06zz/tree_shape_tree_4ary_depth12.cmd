python bench.py --workload tree --tree-shape tree_4ary_depth12 --steps 50 --warmup 10 --blocks 4 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic
