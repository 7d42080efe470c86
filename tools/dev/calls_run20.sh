mkdir -p gpurun_out/r05m
d=/dev/shm/hf; rm -rf $d; mkdir -p $d/reads
flappie_amd/fast5_tool synth $d/reads 32768 3500 5500 20260928 0 1 > /dev/null
python -c "
import sys; sys.path.insert(0,'.')
from flappie_amd import model as M
M.write_mdl('$d/flipflop5_r941native.h', M.synthetic_model(0, 384, seed=1, ident='r941native'))"
for r in 2 4 8 12; do for n in 8192 32768; do
  t0=$(date +%s.%N); env FLAPPIE_MODEL_DIR=$d FLAPPIE_HIP_DEVICE=0 FLAPPIE_CLI_TIMING=1 flappie_amd/flappie --readers $r --limit $n -o $d/out.fq $d/reads 2> $d/err.txt > /dev/null
  t1=$(date +%s.%N); echo "== readers $r limit $n wall $(python -c "print(round($t1-$t0,3))") s"; tail -3 $d/err.txt | cut -c1-160; grep -E "basecalled|wall|fast5 read|signal prep|upload|fetch|write|bound" $d/err.txt | cut -c1-200
done; done > gpurun_out/r05m/hostfed_probe.txt 2>&1
cat gpurun_out/r05m/hostfed_probe.txt
