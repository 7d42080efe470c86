#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/dev/ab_run.sh poll1 poll2 2>&1 | tail -8
cp tools/bin/libffhip_poll2.so flappie_amd/libffhip.so
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_gpu_parity.py tests/test_ragged_gpu.py -m gpu -q -x 2>&1 | tail -4
