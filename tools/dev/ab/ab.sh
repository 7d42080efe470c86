#!/bin/bash
# usage: tools/variants/ab.sh base variant1 variant2 ...   (libffhip_<name>.so under tools/variants; "base" = the tree's library)
cp flappie_amd/libffhip.so /tmp/libffhip_base.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libffhip_base.so flappie_amd/libffhip.so; else cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so; fi
  r=$(timeout 300 python bench.py --config ${CFG:-c2} --steps ${STEPS:-80} --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
except Exception as e: print('failed', e)")
  ok=$(timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -q -m gpu -k "paired" 2>&1 | tail -1 | cut -c1-80)
  echo "$v: $r   [$ok]"
done
done
cp /tmp/libffhip_base.so flappie_amd/libffhip.so
