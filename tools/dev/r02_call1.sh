#!/bin/bash
# first GPU call of round 2: full GPU suite, profiles of four workloads, fuzz seeds 1 and 5 with dumps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rP -x 2>&1 | tail -150 > gpurun_out/r02_call1_pytest.txt
tail -5 gpurun_out/r02_call1_pytest.txt
for c in c2 c4 h256 c5; do
  timeout 900 tools/profile_config.sh r02_$c $c > gpurun_out/r02_prof_$c.log 2>&1
  tail -3 gpurun_out/r02_prof_$c.log | cut -c1-600
done
timeout 400 python tools/dev/diff_fuzz.py 240 1 > gpurun_out/r02_fuzz1.txt 2>&1; tail -3 gpurun_out/r02_fuzz1.txt
timeout 400 python tools/dev/diff_fuzz.py 240 5 > gpurun_out/r02_fuzz5.txt 2>&1; tail -3 gpurun_out/r02_fuzz5.txt
