mkdir -p gpurun_out/r05c
cp flappie_amd/libffhip.so /tmp/tree.so
: > gpurun_out/r05c/diag2.txt
for v in noxpf xpf noxpf xpf; do
cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
echo "== $v" >> gpurun_out/r05c/diag2.txt; python tools/dev/front_order_diag.py 100 FFHIP_DEBUG none >> gpurun_out/r05c/diag2.txt 2>&1
done
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05c/diag2.txt
