#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c2_trace2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c2_trace2 -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-h2d-leg > /dev/null 2>&1
python tools/dev/layer_gaps.py gpurun_out/c2_trace2
rm -rf gpurun_out/c2_trace2
