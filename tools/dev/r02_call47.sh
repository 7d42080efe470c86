#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-h2d-leg --steps 20 --warmup 2 --config c2 --inflight 1"
for d in 0 1 3 7 5 9 17 33 63; do
FFHIP_SKEW=1 FFHIP_SKEW_DBG=$d timeout 300 python bench.py $B | python -c "import json,sys; d=json.load(sys.stdin); print('dbg $d: layer', d['roofline']['avg_launch_ms'], 'ms =', round(d['roofline']['avg_launch_ms']*1e3/800*2400), 'cycles/step')"
done
