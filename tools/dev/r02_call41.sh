#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
timeout 900 python -m pytest tests/test_split_gpu.py -x -q -m gpu > gpurun_out/flaky_$i.log 2>&1; tail -1 gpurun_out/flaky_$i.log; grep -E "^(E  |FAILED)" gpurun_out/flaky_$i.log | head -8
done
