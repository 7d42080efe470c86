#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do timeout 300 python tools/stress.py 40 0 384 2 256 2>&1 | grep -v "^iteration" | tail -3 | cut -c1-300; done
