#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-h2d-leg"
for c in h256 c4; do
  for fl in 1 2; do
  timeout 400 python bench.py --config $c --inflight $fl $B | python -c "import json,sys; d=json.load(sys.stdin); print('$c inflight $fl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
  FFHIP_NO_TAIL_WAIT=1 timeout 400 python bench.py --config $c --inflight 2 $B | python -c "import json,sys; d=json.load(sys.stdin); print('$c inflight 2 no tail wait', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
timeout 300 python -m pytest tests/test_split_gpu.py -x -q -m gpu -k "two_batches or soak or first_ragged" 2>&1 | tail -2
