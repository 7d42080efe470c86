#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/dev/cli_soak.py 4096 2>&1 | tail -12
