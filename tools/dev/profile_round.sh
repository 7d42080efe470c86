#!/bin/bash
# run on the GPU box (via gpurun): kernel trace + the two HBM PMC passes of the default bench, summaries into gpurun_out/
# usage: tools/dev/profile_round.sh TAG
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_trace -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
find gpurun_out/prof_${tag}_* -name "*.csv" | head -20
cat gpurun_out/${tag}_bench.json
