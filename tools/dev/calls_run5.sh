mkdir -p gpurun_out/r05c
cp flappie_amd/libffhip.so /tmp/tree.so
for v in noxpf xpf; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  echo "== $v"; python tools/stress.py 1500 0 384 4 256 1 2>&1 | tail -6
  for k in 1 2 3 4 5 6; do python -m pytest tests/test_bench_shapes_gpu.py -m gpu -q -k front_order 2>&1 | tail -1; done
done > gpurun_out/r05c/stress.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05c/stress.txt
