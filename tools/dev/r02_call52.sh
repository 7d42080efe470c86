#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 60 tools/bin/flow_timeline 16 > gpurun_out/flow_tl.log 2>&1; head -5 gpurun_out/flow_tl.log; grep "blk   0 half-step 20[0-3] [hg]" gpurun_out/flow_tl.log | cut -c1-90
timeout 300 python -m pytest tests/test_split_gpu.py -x -q -m gpu -k "dataflow" 2>&1 | tail -2
FFHIP_FLOW=1 timeout 300 python bench.py --config c2 --inflight 1 --no-cpu-baseline --no-h2d-leg --steps 50 --warmup 3 | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 1 flow 1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
