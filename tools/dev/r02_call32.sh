#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -3
FLAPPIE_NO_READER_THREAD=1 timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -3
for h in 384 256; do
  timeout 1200 python tools/cli_throughput.py $h 2048,8192 4,8 > gpurun_out/r02_cli_h$h.txt 2>&1
  echo "== H $h"; grep "marginal" gpurun_out/r02_cli_h$h.txt; grep -A9 "limit 8192" gpurun_out/r02_cli_h$h.txt | tail -10
done
