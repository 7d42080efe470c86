#!/bin/bash
# CPU seconds a bench.py rank burns while it waits for its GPU: HIP's default (spin) against FFHIP_DEBUG=blocking_sync (run on the GPU box)
TIMEFORMAT="    wall %R s, user %U s, sys %S s (whole process: start-up, model, 400 steps)"
for mode in default blocking default blocking; do
  if [ $mode = blocking ]; then export FFHIP_DEBUG=blocking_sync; else unset FFHIP_DEBUG; fi
  echo "== $mode"
  time python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg --no-length-mix-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('    value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'exposed', d['exposed_ms'])"
done
