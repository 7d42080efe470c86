#!/bin/bash
# tenth GPU call: LDS-staged split convolution against the gather form
cd $GRAFT_REPO_ROOT
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], d["value"], "Msamples/s", d["ms_per_step"], "ms/step; layer", d["roofline"]["avg_launch_ms"], "ms", d["kernel_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "failed", e, open(sys.argv[2].replace(".json", ".err")).read()[-400:])
PY
}
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
for c in c2 h256 c5; do
  FFHIP_CONV_GATHER=1 timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_cg_$c.json 2> gpurun_out/r02_cg_$c.err; show "gather conv $c" gpurun_out/r02_cg_$c.json
  timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_cl_$c.json 2> gpurun_out/r02_cl_$c.err; show "LDS conv    $c" gpurun_out/r02_cl_$c.json
done
