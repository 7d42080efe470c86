mkdir -p gpurun_out/r05c
cp flappie_amd/libffhip.so /tmp/tree.so
cp tools/variants/libffhip_base.so flappie_amd/libffhip.so
echo "== base library (round 4's HEAD) with round 4's test (FFHIP_FRONT_ORDER)" > gpurun_out/r05c/flaky.txt
for k in $(seq 1 16); do python -m pytest tests/_old_front_order_test.py -m gpu -q -k front_order 2>&1 | tail -1; done >> gpurun_out/r05c/flaky.txt
cp /tmp/tree.so flappie_amd/libffhip.so
echo "== tree library with this round's test (FFHIP_DEBUG=front_order=...)" >> gpurun_out/r05c/flaky.txt
for k in $(seq 1 16); do python -m pytest tests/test_bench_shapes_gpu.py -m gpu -q -k front_order 2>&1 | tail -1; done >> gpurun_out/r05c/flaky.txt
cat gpurun_out/r05c/flaky.txt
