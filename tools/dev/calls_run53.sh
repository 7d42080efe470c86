# round 5, call 53: parity campaigns on the final tree (paired H = 384 launch; the packed kernels, whose x waves changed this round), 2048 reads each, against the oracle and the oracle through OpenBLAS
mkdir -p gpurun_out/r05final
for s in c2 h256 c4; do (timeout 1500 python tools/parity_h384.py 2048 2500 $s 2>&1 | tail -25) > gpurun_out/r05final/parity_$s.txt; done
tail -12 gpurun_out/r05final/parity_c2.txt | cut -c1-250; tail -6 gpurun_out/r05final/parity_h256.txt | cut -c1-250; tail -6 gpurun_out/r05final/parity_c4.txt | cut -c1-250
