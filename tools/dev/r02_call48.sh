#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 120 tools/bin/skew_timeline 16 > gpurun_out/skew_tl.log 2>&1; head -21 gpurun_out/skew_tl.log | cut -c1-60,130-230
timeout 600 python -m pytest tests/test_split_gpu.py -x -q -m gpu -k "skewed" 2>&1 | tail -2
FFHIP_SKEW=1 timeout 300 python bench.py --config c2 --inflight 1 --no-cpu-baseline --no-h2d-leg --steps 50 --warmup 3 | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 1 skew', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
