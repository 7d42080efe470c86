mkdir -p gpurun_out/r05j
(python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r05j/suite.txt
for rep in 1 2 3; do for cfg in c2 c4 c5; do for e in "" "no_head_exp"; do
  r=$(FFHIP_DEBUG=$e timeout 300 python bench.py --config $cfg --steps $([ $cfg = c5 ] && echo 4 || echo 40) --warmup 3 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f Msamples/s  %.3f ms/step  launch %.3f  exposed %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('exposed_ms', float('nan'))))")
  echo "$cfg [${e:-head writes E}]: $r"
done; done; done > gpurun_out/r05j/ab.txt 2>&1
cat gpurun_out/r05j/suite.txt gpurun_out/r05j/ab.txt
