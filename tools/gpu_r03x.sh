#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03x
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_differential.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -n 40 $O/pytest.log | cut -c1-400
