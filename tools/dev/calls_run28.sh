# round 5, call 28: k_lstm_pack with piece 1 of the next operand staged in LDS across the gate phase (px4) against px3; rle with the by-kind activation kernel
mkdir -p gpurun_out/r05u; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_px4.so flappie_amd/libffhip.so
(timeout 1200 python -m pytest tests/test_split_gpu.py tests/test_bench_shapes_gpu.py tests/test_gpu_parity.py tests/test_ragged_gpu.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05u/suite.txt
cp /tmp/tree0.so flappie_amd/libffhip.so
CFGS="h256" REPS=4 STEPS=30 tools/dev/ab/multi_ab.sh px3 px4 > gpurun_out/r05u/ab.txt 2>&1
rm -rf /tmp/tr_rle; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_rle -- python bench.py --config rle --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg > gpurun_out/r05u/rle_trace.log 2>&1
f=$(find /tmp/tr_rle -name "*kernel_stats.csv" | head -1); grep -E "k_rle_activate|k_rle_sub|k_rle_partition" $f | cut -c1-60,100-200 > gpurun_out/r05u/rle_kernel_stats.txt
for rep in 1 2 3; do python bench.py --config rle --steps 40 --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rle tree: %.2f Msamples/s  %.3f ms/step  exposed %.3f' % (d['value'], d['ms_per_step'], d.get('exposed_ms', float('nan'))))"; done >> gpurun_out/r05u/ab.txt
cat gpurun_out/r05u/suite.txt gpurun_out/r05u/ab.txt gpurun_out/r05u/rle_kernel_stats.txt
