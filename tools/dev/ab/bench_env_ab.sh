#!/bin/bash
# usage: tools/variants/bench_env_ab.sh "ENV=1" [config]   -- bench with and without an environment switch, interleaved
for rep in 1 2 3; do
for e in "X_UNUSED=1" "$1"; do
  r=$(env $e timeout 300 python bench.py --config ${2:-c2} --steps ${STEPS:-100} --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['exposed_ms'], d['kernel_ms_per_step'])
except Exception as e: print('failed', e)")
  echo "$e: $r"
done
done
