#!/bin/bash
# final profiles of the round + the GPU suite's printed reports
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in c2 h256 c4 c5 rle; do
  timeout 900 tools/profile_config.sh r02_$c $c > gpurun_out/r02_prof3_$c.log 2>&1
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
try:
    d = json.load(open("gpurun_out/profiles/r02_%s_bench.json" % c))
    print(c, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["kernel_ms_per_step"], d["decode_hbm"]["achieved"], d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print(c, "failed", e)
PY
done
timeout 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 > gpurun_out/r02_gpu_suite_full.txt
grep -E "passed|failed" gpurun_out/r02_gpu_suite_full.txt | tail -2
grep -E "fuzz tail|GPU (split|f32)|oracle \(reference|trace cells off|nbase [45]:" gpurun_out/r02_gpu_suite_full.txt > gpurun_out/r02_gpu_suite_reports.txt; cat gpurun_out/r02_gpu_suite_reports.txt
