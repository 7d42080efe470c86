#!/bin/bash
# usage (on the GPU box): CFGS="c2 h256 c4" REPS=2 tools/dev/ab/multi_ab.sh base v1 v2 ...
# bench.py per config with tools/variants/libffhip_<v>.so copied over the tree's library in turn, interleaved; prints value / ms per step / layer launch ms / frac
cp flappie_amd/libffhip.so /tmp/libffhip_tree.so
for cfg in ${CFGS:-c2}; do
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  r=$(timeout 300 python bench.py --config $cfg --steps ${STEPS:-40} --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('%.2f Msamples/s  %.3f ms/step  launch %.3f ms  frac %.4f  exposed %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('exposed_ms', float('nan'))))
except Exception as e: print('failed', e)")
  echo "$cfg $v: $r"
done
done
done
cp /tmp/libffhip_tree.so flappie_amd/libffhip.so
