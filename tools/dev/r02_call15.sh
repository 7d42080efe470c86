#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
for c in c2 c5; do
for v in 1 0; do
if [ $v = 1 ]; then export FFHIP_VITERBI_1P=1; else unset FFHIP_VITERBI_1P; fi
timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_v_$c.json 2> gpurun_out/r02_v_$c.err
python - $c $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/r02_v_%s.json" % sys.argv[1])); print(sys.argv[1], "one-pass" if sys.argv[2] == "1" else "two-pass", d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
done; done
