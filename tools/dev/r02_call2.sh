#!/bin/bash
# second GPU call of round 2: the fp16 two-slice split kernels -- probe, parity suite, accuracy against float64, speed of every workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/bin/f16_denorm_probe > gpurun_out/r02_f16_probe.txt 2>&1; cat gpurun_out/r02_f16_probe.txt
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | tail -120 > gpurun_out/r02_call2_pytest.txt
tail -15 gpurun_out/r02_call2_pytest.txt
for c in c2 h256 c4 c5 rle; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/r02_f16_bench_$c.json 2> gpurun_out/r02_f16_bench_$c.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r02_f16_bench_$c.json"))
    print("$c", d["value"], "Msamples/s", d["ms_per_step"], "ms/step; layer", d["roofline"]["avg_launch_ms"], "ms x", d["roofline"]["launches_per_layer"], "frac", d["roofline"]["frac"], d["kernel_ms_per_step"], "h2d", d.get("h2d_inclusive", {}).get("value"))
except Exception as e:
    print("$c failed", e)
PY
done
timeout 900 python tools/dev/fp64_truth.py 16 2500 256 384 > gpurun_out/r02_f16_fp64_truth.txt 2>&1; cat gpurun_out/r02_f16_fp64_truth.txt
