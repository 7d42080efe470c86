#!/bin/bash
# sixth GPU call: split gate tiles at H = 384 (A/B on one device, interleaved), two batches in flight, full GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
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
  FFHIP_NO_SPLIT_GATE=1 timeout 600 python bench.py --config c2 --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_sg0.json 2> gpurun_out/r02_sg0.err; show "whole tiles  c2" gpurun_out/r02_sg0.json
  timeout 600 python bench.py --config c2 --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_sg1.json 2> gpurun_out/r02_sg1.err; show "split tiles  c2" gpurun_out/r02_sg1.json
done
timeout 600 python bench.py --config c2 --no-cpu-baseline --no-h2d-leg --inflight 2 > gpurun_out/r02_sg1_if2.json 2> gpurun_out/r02_sg1_if2.err; show "inflight 2   c2" gpurun_out/r02_sg1_if2.json
timeout 600 python bench.py --config rle --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_sg1_rle.json 2> gpurun_out/r02_sg1_rle.err; show "split tiles rle" gpurun_out/r02_sg1_rle.json
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
