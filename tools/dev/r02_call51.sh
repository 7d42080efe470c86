#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_split_gpu.py -x -q -m gpu -k "dataflow" 2>&1 | tail -8
B="--no-cpu-baseline --no-h2d-leg --steps 50 --warmup 3"
for fl in 0 1; do
FFHIP_FLOW=$fl timeout 300 python bench.py --config c2 --inflight 1 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 1 flow $fl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
done
FFHIP_FLOW=1 timeout 300 python bench.py --config c2 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 2 flow 1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
