# round 5, call 25: the packed forms' x waves with double-buffered operand pieces (px1: LSTM pack, px2: both packs) against round 4's one piece at a time (px0)
mkdir -p gpurun_out/r05r
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_px2.so flappie_amd/libffhip.so
(timeout 900 python -m pytest tests/test_split_gpu.py tests/test_bench_shapes_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05r/suite.txt
cp /tmp/tree0.so flappie_amd/libffhip.so
CFGS="h256 c4" REPS=3 STEPS=30 tools/dev/ab/multi_ab.sh px0 px1 px2 > gpurun_out/r05r/ab.txt 2>&1
cat gpurun_out/r05r/suite.txt gpurun_out/r05r/ab.txt
