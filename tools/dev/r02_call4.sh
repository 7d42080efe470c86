#!/bin/bash
# fourth GPU call: hardware gate math (FFHIP_FAST_GATES=1) -- speed and accuracy against the exact replay
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], d["value"], "Msamples/s", d["ms_per_step"], "ms/step; layer", d["roofline"]["avg_launch_ms"], "ms", d["kernel_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for c in c2 h256 c4 c5; do
  timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_exact_$c.json 2> gpurun_out/r02_exact_$c.err; show "exact $c" gpurun_out/r02_exact_$c.json
  FFHIP_FAST_GATES=1 timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_fast_$c.json 2> gpurun_out/r02_fast_$c.err; show "fast  $c" gpurun_out/r02_fast_$c.json
done
echo "--- fp64 truth, exact gates"; timeout 900 python tools/dev/fp64_truth.py 16 2500 256 384 2>&1 | tee gpurun_out/r02_truth_exact.txt
echo "--- fp64 truth, hardware gates"; FFHIP_FAST_GATES=1 timeout 900 python tools/dev/fp64_truth.py 16 2500 256 384 2>&1 | tee gpurun_out/r02_truth_fast.txt
echo "--- GPU suite under FFHIP_FAST_GATES=1"
FFHIP_FAST_GATES=1 timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | tail -80 > gpurun_out/r02_call4_pytest_fast.txt
grep -E "passed|failed|^FAILED|fuzz tail|GPU (split|f32)|oracle \(reference|Error" gpurun_out/r02_call4_pytest_fast.txt | head -30
echo "--- parity campaign, exact then hardware gates"
timeout 900 python tools/parity_campaign.py 256 split 2>&1 | tail -3
FFHIP_FAST_GATES=1 timeout 900 python tools/parity_campaign.py 256 split 2>&1 | tail -3
