#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
B="--no-cpu-baseline --no-h2d-leg"
for c in c2 rle; do
  timeout 300 python bench.py --config $c $B | python -c "import json,sys; d=json.load(sys.stdin); print('$c', d['value'], d['ms_per_step'], d['config']['batches_in_flight'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
done
FFHIP_LEAN_CONV=0 timeout 300 python bench.py --config c2 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 fat conv forced', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
FFHIP_LEAN_CONV=1 timeout 300 python bench.py --config c2 --inflight 1 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 1 lean conv forced', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
timeout 300 python bench.py --config c2 --inflight 1 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
