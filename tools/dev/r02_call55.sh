#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
echo "== default rep $rep"; timeout 300 python tools/stress.py 60 0 384 2 256 2>&1 | tail -1 | cut -c1-300
done
echo "== FFHIP_LEAN_CONV=0"; FFHIP_LEAN_CONV=0 timeout 300 python tools/stress.py 90 0 384 2 256 2>&1 | tail -1 | cut -c1-300
echo "== one in flight"; timeout 300 python tools/stress.py 60 0 384 1 256 2>&1 | tail -1 | cut -c1-300
