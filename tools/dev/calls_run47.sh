# round 5, call 47: the LDS-flag variant of the layer kernels (12 deviating batches of 12 000 before) WITH the re-sweep fix: front_order_diag 1000 repeats, then speed against the tree
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_ldsvfix.so flappie_amd/libffhip.so
(timeout 2400 python tools/dev/front_order_diag.py 1000 FFHIP_DEBUG none 2>&1 | cut -c1-200 | tail -6) > gpurun_out/r05z/ldsvfix_diag.txt
cp /tmp/tree0.so flappie_amd/libffhip.so
cp /tmp/tree0.so tools/variants/libffhip_tree.so
CFGS="c2 c4 h256 rle" REPS=3 STEPS=40 tools/dev/ab/multi_ab.sh tree ldsvfix > gpurun_out/r05z/ldsvfix_ab.txt 2>&1
cat gpurun_out/r05z/ldsvfix_diag.txt gpurun_out/r05z/ldsvfix_ab.txt
