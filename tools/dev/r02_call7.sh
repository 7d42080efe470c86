#!/bin/bash
# seventh GPU call: the round's profiles (bench line, kernel trace, HBM and SQ counters) for four workloads; end-to-end CLI throughput by reader count
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in c2 h256 c4 c5; do
  timeout 900 tools/profile_config.sh r02_$c $c > gpurun_out/r02_prof2_$c.log 2>&1
  tail -1 gpurun_out/r02_prof2_$c.log | cut -c1-400
done
timeout 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; cat gpurun_out/r02_bench_default.json | cut -c1-3000
timeout 1200 python tools/cli_throughput.py 384 2048,8192 0,1,2,4,8 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r02_cli_throughput.txt; grep -E "readers|marginal" gpurun_out/r02_cli_throughput.txt
