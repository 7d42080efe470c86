#!/bin/bash
# Run ON THE GPU BOX (through gpurun): every record a round's results table is made of, from the tree as it stands --
#   tools/profile_config.sh TAG_<config> <config> for c2 h256 c4 c5 rle  (bench line, kernel stats, HBM and SQ counter passes, traffic.json)
#   the driver's own command (python bench.py) -> TAG_bench_default.json, and its kernel stats -> TAG_bench_default_kernel_stats.csv
# Everything lands in gpurun_out/profiles/; copy it into profiles/ and run tools/make_tables.py.     usage: tools/profile_all.sh r06 ["c2 h256 c4 c5 rle"]
tag=${1:-r06}; cfgs=${2:-"c2 h256 c4 c5 rle"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/profiles; mkdir -p $O
python bench.py > $O/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || tail -5 gpurun_out/${tag}_bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_default -- python bench.py --no-host-fed-leg --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/prof_${tag}_default -name "*kernel_stats.csv" | xargs ls -S | head -1)
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-host-fed-leg --no-cpu-baseline   (the driver's command, two pairs in flight)"; cat "$f"; } > $O/${tag}_bench_default_kernel_stats.csv
for c in $cfgs; do PROFILE_STEPS=${PROFILE_STEPS:-4} tools/profile_config.sh ${tag}_$c $c > gpurun_out/${tag}_${c}_profile.log 2>&1; tail -1 gpurun_out/${tag}_${c}_profile.log | cut -c1-200; done
ls -la $O
