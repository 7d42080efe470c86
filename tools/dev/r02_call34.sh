#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for h in 384 256; do
  timeout 1200 python tools/cli_throughput.py $h 2048,8192 4,8 > gpurun_out/r02_cli_h$h.txt 2>&1
  echo "== H $h"; grep "marginal" gpurun_out/r02_cli_h$h.txt; grep -A9 "limit 8192" gpurun_out/r02_cli_h$h.txt | tail -10
done
export FLAPPIE_WRAP="rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cli_trace --"
rm -rf gpurun_out/cli_trace
timeout 1200 python tools/cli_throughput.py 256 8192 8 > gpurun_out/r02_cli_trace.txt 2>&1
python tools/dev/trace_gaps.py gpurun_out/cli_trace | head -34
rm -rf gpurun_out/cli_trace
