# round 5, call 33: repeatability of the packed launches -- the tree (px3), px7, px0 (round 4's form)
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
for v in tree px7 px0; do
  [ $v = tree ] && cp /tmp/tree0.so flappie_amd/libffhip.so || cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  echo "== $v"; timeout 900 python tools/dev/pack_repeat.py 150
done > gpurun_out/r05z/repeat.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05z/repeat.txt
