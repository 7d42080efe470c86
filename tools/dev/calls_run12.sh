mkdir -p gpurun_out/r05d
cp flappie_amd/libffhip.so /tmp/tree.so
cp tools/variants/libffhip_xg.so flappie_amd/libffhip.so
(python -m pytest tests/test_split_gpu.py tests/test_bench_shapes_gpu.py tests/test_ragged_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r05d/suite.txt
cp /tmp/tree.so flappie_amd/libffhip.so
CFGS="c2" REPS=3 tools/dev/ab/multi_ab.sh r4like xg xg_p1 xg_p3 xg_s0 xg_s3 > gpurun_out/r05d/ab.txt 2>&1
cp tools/variants/libffhip_xg_phases.so flappie_amd/libffhip.so; python tools/dev/phases.py c2 6 > gpurun_out/r05d/phases.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05d/suite.txt gpurun_out/r05d/ab.txt gpurun_out/r05d/phases.txt
