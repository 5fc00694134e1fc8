python bench.py --workload flat --entities 10000000 --views 1 --steps 50 --warmup 10 --blocks 4 --no-cpu-baseline --no-other-workloads --no-end-to-end --no-live-traffic
