# round 5, call 44: the re-sweep on purpose against a reference from the PLAIN library (the forced path also runs in the one-tile launches the reference came from)
mkdir -p gpurun_out/r05z
cp flappie_amd/libffhip.so /tmp/tree0.so
cp tools/variants/libffhip_px0.so flappie_amd/libffhip.so; timeout 300 python tools/dev/pack_repeat.py 2 GRUmod,LSTM /tmp/ref > /dev/null 2>&1
for v in frB fr0; do
  cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  (echo "== $v (dense, against the plain library's one-tile reference)"; timeout 900 python tools/dev/pack_repeat.py 200 GRUmod,LSTM /tmp/ref) 2>&1 | cut -c1-260 | tail -6
  (echo "== $v, FFHIP_DEBUG=no_dense (one-tile launches, re-sweeping on purpose) against the same reference"; FFHIP_DEBUG=no_dense timeout 900 python tools/dev/pack_repeat.py 60 GRUmod,LSTM /tmp/ref) 2>&1 | cut -c1-260 | tail -4
done > gpurun_out/r05z/diag7.txt 2>&1
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05z/diag7.txt
