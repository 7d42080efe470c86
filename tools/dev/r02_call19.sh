#!/bin/bash
# dense launches at H <= 256: parity, then c4 / h256 at 256, 384 and 512 reads per batch, one and two batches in flight
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_split_gpu.py -x -q -m gpu > gpurun_out/r02_dense_tests.log 2>&1; tail -3 gpurun_out/r02_dense_tests.log
B="--no-cpu-baseline --no-h2d-leg --steps 40 --warmup 3"
for c in c4 h256; do
  for n in 256 384 512; do
    for fl in 1 2; do
      timeout 300 python bench.py --config $c --nread $n --inflight $fl $B > gpurun_out/dense_${c}_${n}_${fl}.json 2>/dev/null
      python - <<PY
import json
d = json.load(open("gpurun_out/dense_${c}_${n}_${fl}.json")); print("$c nread $n inflight $fl", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["kernel_ms_per_step"])
PY
    done
  done
done
FFHIP_NO_DENSE=1 timeout 300 python bench.py --config c4 --nread 512 --inflight 1 $B | python -c "import json,sys; d=json.load(sys.stdin); print('c4 512 NO_DENSE', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
