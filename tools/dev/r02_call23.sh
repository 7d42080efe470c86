#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/dev/inflight_diag.py 2>&1 | grep -v "^   read" | grep -A14 "blocks whose" | head -64
