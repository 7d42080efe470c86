#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in c2 h256 c4 c5 rle; do
  timeout 1200 tools/profile_config.sh r02_$c $c > gpurun_out/r02_prof7_$c.log 2>&1
  python -c "import json; d=json.load(open('gpurun_out/profiles/r02_${c}_bench.json')); print('$c', d['value'], d['ms_per_step'], d['config']['batches_in_flight'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('frac_of_six_product_ceiling'), d['roofline']['traffic'], d['cpu_baseline']['value'], d.get('h2d_inclusive',{}).get('value'), d['decode_hbm']['achieved'])"
  head -5 gpurun_out/profiles/r02_${c}_kernel_stats.csv | tail -3; tail -1 gpurun_out/profiles/r02_${c}_sq_pmc.csv | cut -c1-200
done
timeout 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; python -c "import json; d=json.load(open('gpurun_out/r02_bench_default.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
timeout 600 python bench.py --inflight 1 --no-cpu-baseline > gpurun_out/profiles/r02_c2_inflight1_bench.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/profiles/r02_c2_inflight1_bench.json')); print('inflight1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('frac_of_six_product_ceiling'))"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c2_if2; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2_if2 -- python bench.py --config c2 --steps 4 --warmup 1 --no-cpu-baseline --no-h2d-leg > /dev/null 2>&1
f=$(ls gpurun_out/c2_if2/*/*kernel_stats.csv | head -1)
( echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --config c2 --steps 4 --warmup 1 --no-cpu-baseline --no-h2d-leg   (the default: two batches in flight; durations of co-running kernels overlap)"; head -14 $f ) > gpurun_out/profiles/r02_c2_inflight2_kernel_stats.csv
head -6 gpurun_out/profiles/r02_c2_inflight2_kernel_stats.csv
rm -rf gpurun_out/c2_if2
