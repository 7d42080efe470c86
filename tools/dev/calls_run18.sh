mkdir -p gpurun_out/r05k
cp flappie_amd/libffhip.so /tmp/tree.so
run() { FFHIP_DEBUG=$2 timeout 300 python bench.py --config c2 --steps 40 --warmup 3 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f Msamples/s  %.3f ms/step  launch %.3f  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
cp tools/variants/libffhip_duo.so flappie_amd/libffhip.so
echo "duo: $(run duo '')"; echo "duo no_split_gate: $(run x no_split_gate)"; echo "no_duo: $(run x no_duo)"
cp tools/variants/libffhip_duo_ldscx.so flappie_amd/libffhip.so
echo "duo_ldscx: $(run x '')"
done > gpurun_out/r05k/ab2.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05k/ab2.txt
