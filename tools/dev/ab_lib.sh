#!/bin/bash
# Run ON THE GPU BOX: bench lines of the release library and of a variant library (same C-ABI) in turn.   usage: tools/dev/ab_lib.sh VARIANT.so "c2 h256 c4" [rounds=2]
V=$1; CFGS=${2:-c2}; R=${3:-2}
for cfg in $CFGS; do for r in $(seq $R); do for lib in release $V; do
  if [ $lib = release ]; then unset FFHIP_BINDING_LIBRARY; else export FFHIP_BINDING_LIBRARY=$PWD/$lib; fi
  python bench.py --config $cfg --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', '$lib', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['exposed_ms'])"
done; done; done
