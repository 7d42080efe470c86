#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export FLAPPIE_WRAP="rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cli_trace --"
rm -rf gpurun_out/cli_trace
timeout 1200 python tools/cli_throughput.py 256 8192 8 > gpurun_out/r02_cli_trace.txt 2>&1
tail -12 gpurun_out/r02_cli_trace.txt
python tools/dev/trace_gaps.py gpurun_out/cli_trace
rm -rf gpurun_out/cli_trace
