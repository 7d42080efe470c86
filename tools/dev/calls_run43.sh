# round 5, call 43: where the re-sweeping launch first differs (layer, read tile, block, units)
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_px0.so flappie_amd/libffhip.so; timeout 300 python tools/dev/retry_diff.py dump /tmp/a.npz
cp tools/variants/libffhip_frB.so flappie_amd/libffhip.so; timeout 300 python tools/dev/retry_diff.py dump /tmp/b.npz
cp /tmp/tree0.so flappie_amd/libffhip.so
python tools/dev/retry_diff.py cmp /tmp/a.npz /tmp/b.npz > gpurun_out/r05z/retry_diff.txt 2>&1
head -40 gpurun_out/r05z/retry_diff.txt | cut -c1-250
