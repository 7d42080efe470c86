#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 1500 python tools/parity_campaign.py 256 all inflight; timeout 1500 python tools/parity_campaign.py 256 split inflight ) > gpurun_out/r02_parity_campaign_inflight.txt 2>&1
tail -4 gpurun_out/r02_parity_campaign_inflight.txt
