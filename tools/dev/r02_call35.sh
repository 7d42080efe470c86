#!/bin/bash
cd $GRAFT_REPO_ROOT
for h in 256 384; do
  timeout 1200 python tools/cli_throughput.py $h 2048,8192 8,12 > gpurun_out/r02_cli_h$h.txt 2>&1
  echo "== H $h"; grep "marginal" gpurun_out/r02_cli_h$h.txt; grep -A12 "readers 8 --limit 8192" gpurun_out/r02_cli_h$h.txt | tail -12
done
