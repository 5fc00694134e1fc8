TAG=$1
P=gpurun_out/prof_$TAG
mkdir -p $P
timeout 600 python -m pytest tests -m gpu -q > $P/gputest.log 2>&1; echo "gpu tests rc=$?" >> $P/gputest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $P/smoke.log 2>&1; echo "smoke rc=$?" >> $P/smoke.log
bash tools/profile.sh $TAG > $P/profile.log 2>&1
timeout 900 python bench.py > $P/bench_line.json 2> $P/bench.err; echo "bench rc=$?" >> $P/bench.err
grep -E "passed|failed|rc=" $P/gputest.log $P/smoke.log | tail -5
