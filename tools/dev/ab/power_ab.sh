#!/bin/bash
# usage: tools/dev/ab/power_ab.sh base variant ...  -- the default bench with each library in turn, sampling sclk and socket power under load:
# the layer kernel runs at the 1400 W cap (profiles/r04_power.txt), so the clock it settles at says what a variant costs in energy
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp flappie_amd/libffhip.so /tmp/libffhip_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libffhip_base.so flappie_amd/libffhip.so; else cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so; fi
  env $EXTRA_ENV python bench.py --config ${CFG:-c2} --steps ${STEPS:-1200} --warmup 5 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg > /tmp/pb.json 2>/dev/null &
  pid=$!
  sleep ${SETTLE:-8}
  s=""
  for i in 1 2 3 4; do
    s="$s $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -E 's/.*\(([0-9]+)Mhz\).*/\1MHz/; s/.*Power \(W\): ([0-9.]+).*/\1W/' | tr '\n' ' ')"
    sleep 1
  done
  wait $pid
  r=$(python -c "import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1]); print('%.2f Msamples/s  layer %.3f ms' % (d['value'], d['roofline']['avg_launch_ms']))" 2>/dev/null)
  echo "$v: $r |$s"
done
cp /tmp/libffhip_base.so flappie_amd/libffhip.so
