#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/dev/inflight_diag2.py 2>&1 | tail -12
echo "--- 256 reads"; NREAD=256 timeout 600 python tools/dev/inflight_diag2.py 2>&1 | tail -7
echo "--- batch 0 = H 384, 256 reads"; NREAD=256 HA=384 timeout 600 python tools/dev/inflight_diag2.py 2>&1 | tail -7
