#!/bin/bash
# dense launches (full launches only): split-kernel tests, the 384-read case again, c4 / h256 lines, CLI on an H = 256 model
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_fuzz_tail_gpu.py -x -q -m gpu > gpurun_out/r02_dense_tests.log 2>&1; tail -3 gpurun_out/r02_dense_tests.log
B="--no-cpu-baseline --no-h2d-leg --steps 40 --warmup 3"
for c in c4 h256; do
  for n in 384 768 1024; do
      timeout 300 python bench.py --config $c --nread $n --inflight 1 $B | python -c "import json,sys; d=json.load(sys.stdin); print('$c nread $n', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
  done
done
for c in c4 h256; do
  timeout 600 python bench.py --config $c > gpurun_out/dense_$c.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/dense_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline'], d['kernel_ms_per_step'], d['cpu_baseline']['value'])"
done
timeout 300 python bench.py --config c2 --hidden 128 --nread 512 $B | python -c "import json,sys; d=json.load(sys.stdin); print('H128 nread 512', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
timeout 300 python bench.py --config c2 --hidden 128 --nread 256 $B | python -c "import json,sys; d=json.load(sys.stdin); print('H128 nread 256', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
