mkdir -p gpurun_out/r05c
cp flappie_amd/libffhip.so /tmp/tree.so
echo "== tree" > gpurun_out/r05c/diag.txt; python tools/dev/front_order_diag.py 40 >> gpurun_out/r05c/diag.txt 2>&1
cp tools/variants/libffhip_base.so flappie_amd/libffhip.so
echo "== base" >> gpurun_out/r05c/diag.txt; python tools/dev/front_order_diag.py 40 FFHIP_FRONT_ORDER >> gpurun_out/r05c/diag.txt 2>&1
cp tools/variants/libffhip_ldsv.so flappie_amd/libffhip.so
echo "== ldsv" >> gpurun_out/r05c/diag.txt; python tools/dev/front_order_diag.py 40 FFHIP_FRONT_ORDER >> gpurun_out/r05c/diag.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05c/diag.txt
