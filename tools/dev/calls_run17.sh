mkdir -p gpurun_out/r05k
(timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_split_gpu.py -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r05k/suite.txt
for rep in 1 2 3; do for e in "" "no_duo"; do
  r=$(FFHIP_DEBUG=$e timeout 300 python bench.py --config c2 --steps 40 --warmup 3 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>gpurun_out/r05k/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f Msamples/s  %.3f ms/step  launch %.3f  frac %.4f exposed %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('exposed_ms', float('nan'))))")
  echo "c2 [${e:-duo}]: $r"
done; done > gpurun_out/r05k/ab.txt 2>&1
cat gpurun_out/r05k/suite.txt gpurun_out/r05k/ab.txt; tail -3 gpurun_out/r05k/bench.err
