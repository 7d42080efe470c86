# round 5, call 50: how often the re-sweep path is taken, with LDS flag words (the default) and with the flat accesses
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
for v in count count_flat; do cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so; echo "== $v"; timeout 600 python tools/dev/fallback_count.py 6; done > gpurun_out/r05z/resweep_count.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05z/resweep_count.txt
