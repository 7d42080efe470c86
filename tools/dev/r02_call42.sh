#!/bin/bash
cd $GRAFT_REPO_ROOT
for seed in 7 8 9 10; do
  timeout 400 python tools/dev/diff_fuzz.py 60 $seed > gpurun_out/fuzz_$seed.log 2>&1; echo "seed $seed rc $?"; tail -3 gpurun_out/fuzz_$seed.log | cut -c1-300
done
for i in 1 2 3 4 5 6 7 8; do timeout 100 python tools/dev/diff_fuzz.py 6 7 > gpurun_out/fuzz_s$i.log 2>&1 || { echo "short run $i failed"; tail -5 gpurun_out/fuzz_s$i.log | cut -c1-400; }; done
