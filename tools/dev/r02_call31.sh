#!/bin/bash
cd $GRAFT_REPO_ROOT
for h in 384 256; do
  timeout 1200 python tools/cli_throughput.py $h 2048,8192 4,6,8 > gpurun_out/r02_cli_h$h.txt 2>&1
  echo "== H $h"; grep "marginal" gpurun_out/r02_cli_h$h.txt
done
