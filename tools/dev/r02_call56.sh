#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/stress.py 3000 0 384 2 256 2>&1 | grep -v "^iteration" | tail -12 | cut -c1-400
