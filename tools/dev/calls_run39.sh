# round 5, call 40: d10 = d8 with the x wave held back only until the h wave's hand-off poll has succeeded (not until its recurrent pass is through)
# parity (no flag, never overwritten early), tile A early-released as in px7
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
for v in d10; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  (echo "== $v"; timeout 900 python tools/dev/pack_repeat.py 2000 GRUmod) 2>&1 | cut -c1-300
done > gpurun_out/r05z/diag4.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
grep -E "^==|runs deviate" gpurun_out/r05z/diag4.txt
