# round 5, call 26: packed forms, x pieces of the next step in flight across the gate phase (px3; px5 = both pieces for the LSTM form too, 36 B of scratch) against px2
mkdir -p gpurun_out/r05s
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_px3.so flappie_amd/libffhip.so
(timeout 1200 python -m pytest tests/test_split_gpu.py tests/test_bench_shapes_gpu.py tests/test_gpu_parity.py tests/test_ragged_gpu.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05s/suite.txt
cp /tmp/tree0.so flappie_amd/libffhip.so
CFGS="h256 c4" REPS=3 STEPS=30 tools/dev/ab/multi_ab.sh px2 px3 px5 > gpurun_out/r05s/ab.txt 2>&1
CFGS="rle" REPS=2 STEPS=30 tools/dev/ab/multi_ab.sh px0 px3 >> gpurun_out/r05s/ab.txt 2>&1
cat gpurun_out/r05s/suite.txt gpurun_out/r05s/ab.txt
