#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_suite.log 2>&1; tail -2 gpurun_out/r02_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
