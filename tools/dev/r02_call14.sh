#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in c2 c5; do
  timeout 900 tools/profile_config.sh r02_$c $c > gpurun_out/r02_prof4_$c.log 2>&1
done
for c in h256 c4 rle; do
  timeout 600 python bench.py --config $c > gpurun_out/profiles/r02_${c}_bench.json 2> gpurun_out/r02_${c}_bench.err
done
timeout 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
python - <<'PY'
import json
for c in ("c2", "h256", "c4", "c5", "rle"):
    d = json.load(open("gpurun_out/profiles/r02_%s_bench.json" % c))
    print(c, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["kernel_ms_per_step"], d["decode_hbm"]["achieved"], d.get("cpu_baseline", {}).get("value"), d.get("h2d_inclusive", {}).get("value"))
d = json.load(open("gpurun_out/r02_bench_default.json")); print("default", d["value"], d["ms_per_step"], d["roofline"], d["cpu_baseline"]["value"], d["cpu_baseline"]["one_core_alone"])
PY
