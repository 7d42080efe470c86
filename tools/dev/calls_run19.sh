mkdir -p gpurun_out/r05l gpurun_out/profiles
(python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r05l/suite.txt
python bench.py > gpurun_out/profiles/r05_bench_default.json 2> gpurun_out/r05l/bench_default.err
for c in h256 c4 c5 rle; do python bench.py --config $c --no-cpu-baseline --no-host-fed-leg > gpurun_out/profiles/r05_${c}_bench.json 2>/dev/null; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_default2 -- python bench.py --no-cpu-baseline --no-host-fed-leg --no-h2d-leg > /dev/null 2>&1
f=$(find gpurun_out/prof_r05_default2 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/profiles/r05_bench_default_kernel_stats.csv
cat gpurun_out/r05l/suite.txt; for c in bench_default h256_bench c4_bench c5_bench rle_bench; do python -c "
import json,sys; d=json.load(open('gpurun_out/profiles/r05_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('exposed_ms'), d.get('decode_hbm',{}).get('achieved'), (d.get('host_fed') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value'))"; done; head -8 gpurun_out/profiles/r05_bench_default_kernel_stats.csv | cut -c1-160
