#!/bin/bash
# Run ON THE GPU BOX: bench lines of the release library and of several variant libraries in turn.   usage: tools/dev/ab_multi.sh "c2" 2 VARIANT1.so VARIANT2.so ...
CFGS=$1; R=$2; shift; shift
for cfg in $CFGS; do for r in $(seq $R); do for lib in release "$@"; do
  if [ $lib = release ]; then unset FFHIP_BINDING_LIBRARY; else export FFHIP_BINDING_LIBRARY=$PWD/$lib; fi
  python bench.py --config $cfg --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', '$lib', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done; done; done
