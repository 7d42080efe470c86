#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -1; done
timeout 1500 python tools/parity_campaign.py 256 all inflight 2>&1 | tail -1
for i in 1 2 3 4; do timeout 200 python tools/dev/diff_fuzz.py 8 $i 2>&1 | tail -1 | cut -c1-200; done
