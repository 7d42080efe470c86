# round 5, call 24: a layer launch ALONE on the chip (one 256-read batch at a time): phases of the dense form, the one-tile form and the 256-register pair form
mkdir -p gpurun_out/r05q
for rep in 1 2; do for v in "no_pair" "split_ts=2" "split_ts=1"; do
  r=$(FFHIP_DEBUG="$v" timeout 300 python bench.py --config c2 --no-pair --inflight 1 --steps 30 --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>gpurun_out/r05q/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f Msamples/s  %.3f ms/step  launch %.3f ms x %.1f per layer  kernel %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches_per_layer'], d['roofline']['kernel'][:40]))")
  echo "c2 one batch at a time [${v:-default}]: $r"
done; done > gpurun_out/r05q/forms.txt 2>&1
cp flappie_amd/libffhip.so /tmp/tree.so; cp tools/variants/libffhip_phases.so flappie_amd/libffhip.so
for v in "no_pair" "split_ts=2" "split_ts=1"; do echo "== phases, serial [${v:-default}]"; FFHIP_DEBUG="$v" timeout 300 python tools/dev/phases.py c2 4 serial; done > gpurun_out/r05q/phases.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05q/forms.txt gpurun_out/r05q/phases.txt; tail -3 gpurun_out/r05q/err.txt
