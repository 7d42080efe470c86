#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 8 16 24; do
echo "== FFHIP_SKEW_DBG=$d"
FFHIP_SKEW_DBG=$d timeout 120 tools/bin/skew_timeline 16 > gpurun_out/skew_tl_$d.log 2>&1; grep "rep 29" gpurun_out/skew_tl_$d.log; grep "blk   0 half-step 20[12]" gpurun_out/skew_tl_$d.log | cut -c1-30,125-230
done
