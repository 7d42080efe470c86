# round 5, call 52: phases of the final kernels (LDS flag words, re-sweep fix, packed x pieces): c2, h256, c4
mkdir -p gpurun_out/r05final
cp flappie_amd/libffhip.so /tmp/tree0.so; cp tools/variants/libffhip_phases.so flappie_amd/libffhip.so
for c in c2 h256 c4; do timeout 300 python tools/dev/phases.py $c 4; done > gpurun_out/r05final/phases.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05final/phases.txt
