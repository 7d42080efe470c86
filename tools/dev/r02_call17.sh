#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in c5 rle c2; do
for f in 1 2; do
timeout 900 python bench.py --config $c --no-cpu-baseline --no-h2d-leg --inflight $f > gpurun_out/r02_if_$c.json 2> gpurun_out/r02_if_$c.err
python - $c $f <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/r02_if_%s.json" % sys.argv[1])); print(sys.argv[1], "inflight", sys.argv[2], d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
except Exception as e:
    print(sys.argv[1], sys.argv[2], "failed", open("gpurun_out/r02_if_%s.err" % sys.argv[1]).read()[-300:])
PY
done; done
