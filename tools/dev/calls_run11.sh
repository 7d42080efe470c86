mkdir -p gpurun_out/r05c
cp flappie_amd/libffhip.so /tmp/tree.so
: > gpurun_out/r05c/diag5.txt
for v in f7 f1 f2 f4; do
cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
echo "== $v" >> gpurun_out/r05c/diag5.txt; python tools/dev/front_order_diag.py 400 FFHIP_DEBUG none 2>&1 | cut -c1-130 >> gpurun_out/r05c/diag5.txt
done
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05c/diag5.txt
