#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 900 python tools/stress.py 600 0 512 2 256 2>&1 | grep -v "^iteration" | tail -10
  timeout 900 python tools/stress.py 600 1 256 2 512 2>&1 | grep -v "^iteration" | tail -10
  timeout 900 python tools/stress.py 300 0 512 1 256 2>&1 | grep -v "^iteration" | tail -4
  timeout 900 python tools/stress.py 300 1 256 1 512 2>&1 | grep -v "^iteration" | tail -4 ) > gpurun_out/r02_stress.txt 2>&1
cut -c1-400 gpurun_out/r02_stress.txt
