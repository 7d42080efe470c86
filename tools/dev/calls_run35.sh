# round 5, call 35: failure statistics -- px7 3000 runs, the tree (px3) 10 000, px0 (round 4's form) 5000, per cell kind
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_px7.so flappie_amd/libffhip.so; (echo "== px7"; timeout 1500 python tools/dev/pack_repeat.py 3000) > gpurun_out/r05z/repeat_px7_3000.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so; (echo "== tree (px3)"; timeout 2400 python tools/dev/pack_repeat.py 10000) > gpurun_out/r05z/repeat_tree_10000.txt 2>&1
cp tools/variants/libffhip_px0.so flappie_amd/libffhip.so; (echo "== px0"; timeout 1500 python tools/dev/pack_repeat.py 5000) > gpurun_out/r05z/repeat_px0_5000.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cut -c1-400 gpurun_out/r05z/repeat_px7_3000.txt | tail -30; tail -3 gpurun_out/r05z/repeat_tree_10000.txt gpurun_out/r05z/repeat_px0_5000.txt
