set -x
mkdir -p gpurun_out/prof2
export TMPDIR=/tmp
P=gpurun_out/prof2
for wl in flat tree lights; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$wl -o $wl -- python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline > $P/$wl.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/flat_fetch -o flat -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $P/flat_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/flat_write -o flat -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $P/flat_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $P/flat_tcc -o flat -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $P/flat_tcc.log 2>&1
find $P -type f | head -50
du -sh $P
