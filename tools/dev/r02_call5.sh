#!/bin/bash
# fifth GPU call: one read tile per group / two workgroups per CU (FFHIP_SPLIT_TS=1) against the pair form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], d["value"], "Msamples/s", d["ms_per_step"], "ms/step; layer", d["roofline"]["avg_launch_ms"], "ms x", d["roofline"]["launches_per_layer"])
except Exception as e:
    print(sys.argv[1], "failed", e, open(sys.argv[2].replace(".json", ".err")).read()[-400:])
PY
}
for c in h256 c4 c2 c5; do
  for ts in 2 1; do
    FFHIP_SPLIT_TS=$ts timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_ts${ts}_$c.json 2> gpurun_out/r02_ts${ts}_$c.err; show "TS=$ts $c" gpurun_out/r02_ts${ts}_$c.json
  done
done
echo "--- split / ragged / parity tests with one tile per group"
FFHIP_SPLIT_TS=1 timeout 1200 python -m pytest tests/test_split_gpu.py tests/test_ragged_gpu.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -15
