# round 5, call 27: px0 / px2 / px3 once more on another box; phases of the packed forms with px3; the run-length model's serial kernel stats with the 2-D element-wise kernels
mkdir -p gpurun_out/r05t; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFGS="h256 c4" REPS=3 STEPS=30 tools/dev/ab/multi_ab.sh px0 px2 px3 > gpurun_out/r05t/ab.txt 2>&1
cp flappie_amd/libffhip.so /tmp/tree.so; cp tools/variants/libffhip_phases.so flappie_amd/libffhip.so
for c in h256 c4; do timeout 300 python tools/dev/phases.py $c 4; done > gpurun_out/r05t/phases.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
rm -rf /tmp/tr_rle; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_rle -- python bench.py --config rle --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg > gpurun_out/r05t/rle_trace.log 2>&1
python tools/profile_summary.py /tmp/tr_rle > gpurun_out/r05t/rle_kernel_stats.txt 2>&1 || (f=$(find /tmp/tr_rle -name "*kernel_stats.csv" | head -1); head -20 $f > gpurun_out/r05t/rle_kernel_stats.txt)
cat gpurun_out/r05t/ab.txt gpurun_out/r05t/phases.txt; head -20 gpurun_out/r05t/rle_kernel_stats.txt
