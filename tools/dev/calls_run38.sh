# round 5, call 38: what about the early release fails?  on px7 (34 of 3000):  d5 = the x wave's write of the partials delayed ~2000 cycles behind the flag;
# d6 = the flag as ds_write / ds_read instead of flat accesses;  d7 = early release for the first tile, late release (a third flag word) for the second
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
for v in d5 d6 d7; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  (echo "== $v"; timeout 900 python tools/dev/pack_repeat.py 2000 GRUmod) 2>&1 | cut -c1-300
done > gpurun_out/r05z/diag2.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
grep -E "^==|runs deviate" gpurun_out/r05z/diag2.txt
