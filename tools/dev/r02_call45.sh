#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_split_gpu.py -x -q -m gpu -k "skewed" 2>&1 | tail -12
B="--no-cpu-baseline --no-h2d-leg --steps 50 --warmup 3"
for sk in 0 1; do
FFHIP_SKEW=$sk timeout 300 python bench.py --config c2 --inflight 1 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 1 skew $sk', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
done
FFHIP_SKEW=1 timeout 300 python bench.py --config c2 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 2 skew 1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
FFHIP_SKEW=1 timeout 300 python bench.py --config c5 --no-cpu-baseline --no-h2d-leg | python -c "import json,sys; d=json.load(sys.stdin); print('c5 skew 1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
