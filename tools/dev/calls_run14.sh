mkdir -p gpurun_out/r05g
(python -m pytest tests/test_split_gpu.py tests/test_bench_shapes_gpu.py tests/test_ragged_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05g/suite.txt
CFGS="c2 rle" REPS=4 tools/dev/ab/multi_ab.sh sgo0 sgo > gpurun_out/r05g/ab.txt 2>&1
cat gpurun_out/r05g/suite.txt gpurun_out/r05g/ab.txt
