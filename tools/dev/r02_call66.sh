#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 2400 python tools/parity_campaign.py 512 all inflight 2>&1 | tail -1
  timeout 2400 python tools/parity_campaign.py 512 split inflight 2>&1 | tail -1
  for seed in 11 12 13; do timeout 300 python tools/dev/diff_fuzz.py 60 $seed 2>&1 | tail -1; done ) > gpurun_out/r02_parity_campaign_final.txt 2>&1
cut -c1-300 gpurun_out/r02_parity_campaign_final.txt
