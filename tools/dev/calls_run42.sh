# round 5, call 42: round 4's form with (frA) one member in eight ~2500 cycles late every 32nd step, nothing re-swept; (frB) the re-sweep taken by ALL members at the same steps
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
for v in frA frB; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  (echo "== $v"; timeout 900 python tools/dev/pack_repeat.py 300) 2>&1 | cut -c1-200 | grep -E "^==|runs deviate"
done > gpurun_out/r05z/diag6.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05z/diag6.txt
