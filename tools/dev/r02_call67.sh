#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python tools/dev/trace_cost.py 2048 > gpurun_out/trace_cost_a.txt 2>&1
cut -c1-620 gpurun_out/trace_cost_a.txt | tail -9
timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -1
