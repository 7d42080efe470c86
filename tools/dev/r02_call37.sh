#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for h in 256 384; do
  timeout 1500 python tools/cli_throughput.py $h 8192,32768 8,12 > gpurun_out/r02_cli_h$h.txt 2>&1
  echo "== H $h"; grep "marginal" gpurun_out/r02_cli_h$h.txt; grep -A13 "readers 8 --limit 32768" gpurun_out/r02_cli_h$h.txt | tail -13
done
