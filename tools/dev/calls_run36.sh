# round 5, call 36: which change to px7 (the variant that fails ~1 launch in 150) makes the failure go away?  GRUmod only, 3000 launches each
#  d1: every wait of the sweep's landing is vmcnt(0)   d2: sentinels six steps ahead instead of three   d3: the projection partials released behind the recurrent pass
#  d4: vmcnt(0) + lgkmcnt(0) before the h waves' closing barrier
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
for v in px7 d1 d2 d3 d4; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  (echo "== $v"; timeout 900 python tools/dev/pack_repeat.py 3000 GRUmod) 2>&1 | cut -c1-400
done > gpurun_out/r05z/diag.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
grep -E "^==|runs deviate" gpurun_out/r05z/diag.txt
