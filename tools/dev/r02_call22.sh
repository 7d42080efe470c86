#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_split_gpu.py -x -q -m gpu -k "two_batches or dense" > gpurun_out/r02_inflight_tests.log 2>&1; tail -5 gpurun_out/r02_inflight_tests.log
