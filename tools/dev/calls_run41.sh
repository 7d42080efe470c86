# round 5, call 41: is it the RE-SWEEP path?  (a sweep that meets a sentinel is done again; a parked x wave lets the h wave sweep at the earliest instant, so that path runs more often)
# fr3 = the cured form (px3 + late release) with the re-sweep path taken on purpose every 32nd step; fr0 = round 4's form with the same; both kinds
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
for v in fr3 fr0; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  (echo "== $v"; timeout 900 python tools/dev/pack_repeat.py 300) 2>&1 | cut -c1-300 | tail -8
done > gpurun_out/r05z/diag5.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05z/diag5.txt
