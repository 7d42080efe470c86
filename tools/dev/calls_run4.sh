mkdir -p gpurun_out/r05c
cp flappie_amd/libffhip.so /tmp/tree.so
for v in noxpf xpf noxpf xpf; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  echo "== $v" ; python -m pytest tests/test_bench_shapes_gpu.py -m gpu -q -k front_order 2>&1 | tail -3
done > gpurun_out/r05c/front_order.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
CFGS="c2" REPS=3 tools/dev/ab/multi_ab.sh noxpf xpf xp1 xp2 xp3 noxpf_xp3 > gpurun_out/r05c/ab.txt 2>&1
cat gpurun_out/r05c/front_order.txt gpurun_out/r05c/ab.txt
