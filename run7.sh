export TMPDIR=/tmp
for b in 256 512 1024; do
  echo "== MI_TILE_BLOCK=$b"
  MI_TILE_BLOCK=$b timeout 300 python bench.py --no-cpu-baseline --workload tree --steps 100 --profile-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels'], d['roofline']['launches'])"
done
MI_TILE_BLOCK=512 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r7/tree -o tree -- python bench.py --workload tree --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cat gpurun_out/r7/tree/tree_kernel_stats.csv
