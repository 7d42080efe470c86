#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
( timeout 1500 python tools/stress.py 3000 0 384 2 256 2>&1 | tail -1
  timeout 900 python tools/stress.py 800 0 256 2 512 2>&1 | tail -1
  timeout 900 python tools/stress.py 600 1 256 2 512 2>&1 | tail -1
  timeout 900 python tools/stress.py 400 0 512 2 256 2>&1 | tail -1 ) > gpurun_out/r02_stress.txt 2>&1
cat gpurun_out/r02_stress.txt
