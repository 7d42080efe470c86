#!/bin/bash
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -3
( timeout 600 python tools/stress.py 300 0 384 2 256 2>&1 | grep -v "^iteration" | tail -6
  timeout 600 python tools/stress.py 150 1 256 2 512 2>&1 | grep -v "^iteration" | tail -6
  timeout 600 python tools/stress.py 100 0 512 2 256 2>&1 | grep -v "^iteration" | tail -6
  timeout 600 python tools/stress.py 200 0 256 2 512 2>&1 | grep -v "^iteration" | tail -6 ) 2>&1 | cut -c1-420
