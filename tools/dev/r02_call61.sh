#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_suite.log 2>&1; tail -2 gpurun_out/r02_gpu_suite.log
timeout 600 python bench.py > gpurun_out/r02_bench_default.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r02_bench_default.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['cpu_baseline']['value'], d['h2d_inclusive']['value'])"
