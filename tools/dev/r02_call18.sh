#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in c2 rle; do
  timeout 900 tools/profile_config.sh r02_$c $c > gpurun_out/r02_prof5_$c.log 2>&1
done
timeout 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
timeout 600 python bench.py --inflight 1 --no-cpu-baseline > gpurun_out/profiles/r02_c2_inflight1_bench.json 2> /dev/null
python - <<'PY'
import json
for f in ("gpurun_out/profiles/r02_c2_bench.json", "gpurun_out/profiles/r02_rle_bench.json", "gpurun_out/r02_bench_default.json", "gpurun_out/profiles/r02_c2_inflight1_bench.json"):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["batches_in_flight"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"].get("frac_of_six_product_ceiling"), d["kernel_ms_per_step"], d.get("h2d_inclusive", {}).get("value"), d.get("cpu_baseline", {}).get("value"))
PY
head -9 gpurun_out/profiles/r02_c2_kernel_stats.csv; tail -1 gpurun_out/profiles/r02_c2_sq_pmc.csv
