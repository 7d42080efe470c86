#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_gpu_parity.py tests/test_fuzz_tail_gpu.py -x -q -m gpu 2>&1 | tail -2
B="--no-cpu-baseline --no-h2d-leg"
for c in c2 h256 c4 c5 rle; do
  timeout 400 python bench.py --config $c $B | python -c "import json,sys; d=json.load(sys.stdin); print('$c', d['value'], d['ms_per_step'], d['config']['batches_in_flight'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
timeout 300 python bench.py --config c2 --inflight 1 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['frac_of_six_product_ceiling'])"
