# round 5, call 29: phases of the H = 512 form (k_lstm_split<0,4,2>, one workgroup a CU, 256 registers; c5's kernel) with two batches in flight and alone
mkdir -p gpurun_out/r05v
cp flappie_amd/libffhip.so /tmp/tree.so; cp tools/variants/libffhip_phases.so flappie_amd/libffhip.so
(echo "== h512, two batches submitted"; timeout 600 python tools/dev/phases.py h512 2; echo "== h512, one batch at a time"; timeout 600 python tools/dev/phases.py h512 2 serial) > gpurun_out/r05v/phases.txt 2>&1
cp /tmp/tree.so flappie_amd/libffhip.so
cat gpurun_out/r05v/phases.txt
