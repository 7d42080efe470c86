#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in c4 h256; do
  timeout 900 tools/profile_config.sh r02_$c $c > gpurun_out/r02_prof6_$c.log 2>&1
  python -c "import json; d=json.load(open('gpurun_out/profiles/r02_${c}_bench.json')); print('$c', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d.get('h2d_inclusive',{}).get('value'))"
  head -6 gpurun_out/profiles/r02_${c}_kernel_stats.csv; tail -2 gpurun_out/profiles/r02_${c}_sq_pmc.csv
done
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_suite.log 2>&1; tail -3 gpurun_out/r02_gpu_suite.log
