#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_suite.log 2>&1; tail -2 gpurun_out/r02_gpu_suite.log
for h in 384 256; do
  timeout 1500 python tools/cli_throughput.py $h 8192,32768 8,12 > gpurun_out/r02_cli_h$h.txt 2>&1
  echo "== H $h"; grep "marginal" gpurun_out/r02_cli_h$h.txt
done
