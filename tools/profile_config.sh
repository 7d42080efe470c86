#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything behind one bench line of one workload --
#   1. the bench line itself (default steps)                         -> gpurun_out/profiles/TAG_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command          -> TAG_kernel_stats.csv
#   3. HBM traffic: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes -> TAG_hbm_traffic_pmc.csv (+ TAG_traffic.json entry)
#   4. matrix-pipe counters of the recurrent kernel, own pass(es)    -> TAG_sq_pmc.csv
# Copy gpurun_out/profiles/TAG_* into profiles/ afterwards (gpurun_out/ is scratch).
# usage: tools/profile_config.sh TAG CONFIG [extra bench.py flags]
tag=$1; cfg=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/profiles; mkdir -p $O
S="--config $cfg --steps ${PROFILE_STEPS:-4} --warmup 2 --inflight 1 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg $*"      # one batch in flight: clean per-kernel times
python bench.py --config $cfg --no-host-fed-leg "$@" > $O/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || tail -5 gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_trace -- python bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${tag}_fetch -- python bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${tag}_write -- python bench.py $S > /dev/null 2>&1
# SQ counters in passes small enough for the hardware's counter slots
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_${tag}_sq1 -- python bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_${tag}_sq2 -- python bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/prof_${tag}_sq3 -- python bench.py $S > /dev/null 2>&1
python tools/profile_summary.py $tag $cfg gpurun_out $O
cat $O/${tag}_bench.json
