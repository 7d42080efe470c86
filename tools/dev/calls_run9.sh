mkdir -p gpurun_out/r05c
cp flappie_amd/libffhip.so /tmp/tree.so
: > gpurun_out/r05c/diag3.txt
for v in base ldsv; do
cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
echo "== $v" >> gpurun_out/r05c/diag3.txt; python tools/dev/front_order_diag.py 300 FFHIP_FRONT_ORDER none 2>&1 | cut -c1-200 >> gpurun_out/r05c/diag3.txt
done
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05c/diag3.txt
