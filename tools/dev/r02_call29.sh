#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/check_isa.py
timeout 600 python tools/dev/inflight_diag2.py 2>&1 | tail -7
echo "--- 256 reads, batch 0 = H 384"; NREAD=256 HA=384 timeout 600 python tools/dev/inflight_diag2.py 2>&1 | tail -7
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_suite.log 2>&1; tail -3 gpurun_out/r02_gpu_suite.log
B="--no-cpu-baseline --no-h2d-leg"
for c in c2 h256 c4 rle; do
  timeout 300 python bench.py --config $c $B | python -c "import json,sys; d=json.load(sys.stdin); print('$c', d['value'], d['ms_per_step'], d['config']['batches_in_flight'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
done
timeout 300 python bench.py --config c2 --inflight 1 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c2 inflight 1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
