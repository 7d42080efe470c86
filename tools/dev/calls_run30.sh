# round 5, call 30: H = 512 form, the x waves' prefetch inside the projection (xi1) against behind it (xi0): c5, parity at H = 512, phases
mkdir -p gpurun_out/r05w
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_xi1.so flappie_amd/libffhip.so
(timeout 1500 python -m pytest tests/test_bench_shapes_gpu.py tests/test_long_reads_gpu.py tests/test_split_gpu.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05w/suite.txt
cp /tmp/tree0.so flappie_amd/libffhip.so
CFGS="c5" REPS=3 STEPS=4 tools/dev/ab/multi_ab.sh xi0 xi1 > gpurun_out/r05w/ab.txt 2>&1
cp tools/variants/libffhip_phases.so flappie_amd/libffhip.so
(echo "== h512 phases, xi1"; timeout 600 python tools/dev/phases.py h512 2) > gpurun_out/r05w/phases.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05w/suite.txt gpurun_out/r05w/ab.txt gpurun_out/r05w/phases.txt
