#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03u
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -n 12 $O/pytest.log
