#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in c4 h256 c2; do
for f in 1 2; do
timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg --inflight $f > gpurun_out/r02_if_$c.json 2> gpurun_out/r02_if_$c.err
python - $c $f <<'PY'
import json, sys
d = json.load(open("gpurun_out/r02_if_%s.json" % sys.argv[1])); print(sys.argv[1], "inflight", sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
done; done
