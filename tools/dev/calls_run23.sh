# round 5, call 23: the layer kernel's OTHER forms at c2's shape, one batch in flight -- what a step costs with one workgroup a CU and a pair of tiles (split_ts=2, 256 registers)
mkdir -p gpurun_out/r05p
for rep in 1 2; do for v in "" "no_pair" "split_ts=2" "split_ts=1"; do
  r=$(FFHIP_DEBUG="$v" timeout 300 python bench.py --config c2 --inflight 1 --steps 30 --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f Msamples/s  %.3f ms/step  launch %.3f ms x %.1f per layer  kernel %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches_per_layer'], d['roofline']['kernel'][:40]))")
  echo "c2 inflight 1 [${v:-default}]: $r"
done; done > gpurun_out/r05p/forms.txt 2>&1
cp flappie_amd/libffhip.so /tmp/tree.so; cp tools/variants/libffhip_phases.so flappie_amd/libffhip.so
for v in "" "split_ts=2"; do echo "== phases [${v:-default}]"; FFHIP_DEBUG="$v" timeout 300 python tools/dev/phases.py c2 4; done > gpurun_out/r05p/phases.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05p/forms.txt gpurun_out/r05p/phases.txt
