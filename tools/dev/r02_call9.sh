#!/bin/bash
# ninth GPU call: one-tile form at N = 3 with K in thirds (6-wave workgroups, two per CU)
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
for round in 1 2; do
  for ts in 2 1; do
    FFHIP_SPLIT_TS=$ts timeout 600 python bench.py --config c2 --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_kw_ts$ts.json 2> gpurun_out/r02_kw_ts$ts.err; show "TS=$ts c2" gpurun_out/r02_kw_ts$ts.json
  done
done
FFHIP_SPLIT_TS=1 timeout 1200 python -m pytest tests/test_split_gpu.py tests/test_ragged_gpu.py tests/test_gpu_parity.py tests/test_fuzz_tail_gpu.py -m gpu -q 2>&1 | tail -8
