python bench.py --workload flat_static --entities 10000000 --views 4 --steps 50 --warmup 10 --blocks 4 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic
