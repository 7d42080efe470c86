# round 5, call 31: packed forms, the cross-step pieces issued at the TOP OF THE GATE PHASE (px6; px7: both pieces for the LSTM form too) against behind the projection (px3)
mkdir -p gpurun_out/r05x
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_px7.so flappie_amd/libffhip.so
(timeout 1200 python -m pytest tests/test_split_gpu.py tests/test_bench_shapes_gpu.py tests/test_gpu_parity.py tests/test_ragged_gpu.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05x/suite.txt
cp /tmp/tree0.so flappie_amd/libffhip.so
CFGS="h256 c4" REPS=3 STEPS=30 tools/dev/ab/multi_ab.sh px3 px6 px7 > gpurun_out/r05x/ab.txt 2>&1
cat gpurun_out/r05x/suite.txt gpurun_out/r05x/ab.txt
