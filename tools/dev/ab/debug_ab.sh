#!/bin/bash
# usage (on the GPU box): CFGS="c2 h256 c4" REPS=2 tools/dev/ab/debug_ab.sh "" "front_order=head" ...
# bench.py per config with FFHIP_DEBUG set to each argument in turn ("" = the default), interleaved; prints value / ms per step / layer launch ms / frac / exposed
for cfg in ${CFGS:-c2}; do
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  r=$(FFHIP_DEBUG="$v" timeout 300 python bench.py --config $cfg --steps ${STEPS:-40} --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('%.2f Msamples/s  %.3f ms/step  launch %.3f ms  frac %.4f  exposed %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('exposed_ms', float('nan'))))
except Exception as e: print('failed', e)")
  echo "$cfg [${v:-default}]: $r"
done
done
done
