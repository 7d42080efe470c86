mkdir -p gpurun_out/r05f
tools/variants/coissue_probe 2>&1 | grep -E "older|^M \+ M|^M alone" > gpurun_out/r05f/coissue3.txt
cp flappie_amd/libffhip.so /tmp/tree.so
cp tools/variants/libffhip_swap.so flappie_amd/libffhip.so
(python -m pytest tests/test_split_gpu.py tests/test_bench_shapes_gpu.py tests/test_ragged_gpu.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05f/suite.txt
cp /tmp/tree.so flappie_amd/libffhip.so
CFGS="c2 h256 c4" REPS=3 tools/dev/ab/multi_ab.sh r4like swap > gpurun_out/r05f/ab.txt 2>&1
cat gpurun_out/r05f/coissue3.txt gpurun_out/r05f/suite.txt gpurun_out/r05f/ab.txt
