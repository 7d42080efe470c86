# round 5, call 46: the re-sweep regression test on the tree (fixed) and on frB (round 4's kernels re-sweeping on purpose, no fix: must FAIL); then the whole GPU suite; then 4000 plain launches
mkdir -p gpurun_out/r05z
(timeout 1500 python -m pytest tests/test_resweep_gpu.py -m gpu -q 2>&1 | tail -4) > gpurun_out/r05z/resweep_test.txt
(echo "--- the same test against a library WITHOUT the fix (tools/variants/libffhip_frB.so):"; FFHIP_TEST_RESWEEP_LIB=$PWD/tools/variants/libffhip_frB.so timeout 1500 python -m pytest tests/test_resweep_gpu.py -m gpu -q 2>&1 | tail -9 | cut -c1-200) >> gpurun_out/r05z/resweep_test.txt
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r05z/suite.txt
timeout 900 python tools/dev/pack_repeat.py 4000 GRUmod > gpurun_out/r05z/repeat_fixed_4000.txt 2>&1
cat gpurun_out/r05z/resweep_test.txt gpurun_out/r05z/suite.txt; tail -2 gpurun_out/r05z/repeat_fixed_4000.txt
