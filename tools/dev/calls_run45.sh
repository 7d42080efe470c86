# round 5, call 45: the re-sweep on purpose with vmcnt(0) + lgkmcnt(0) in front of every re-sweep (frC: one member in eight, frD: all members), against the plain library's reference
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_px0.so flappie_amd/libffhip.so; timeout 300 python tools/dev/pack_repeat.py 2 GRUmod,LSTM /tmp/ref > /dev/null 2>&1
for v in frC frD frB; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  (echo "== $v"; timeout 900 python tools/dev/pack_repeat.py 200 GRUmod,LSTM /tmp/ref) 2>&1 | cut -c1-200 | grep -E "^==|runs deviate"
done > gpurun_out/r05z/diag8.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05z/diag8.txt
