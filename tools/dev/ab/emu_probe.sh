#!/bin/bash
# probe: where does an emulated-GPU flappie process spend its time?
d=/dev/shm/emu_probe; rm -rf $d; mkdir -p $d/reads
flappie_amd/fast5_tool synth $d/reads 16384 3500 5500 20260928 0 1 > /dev/null
python - <<PY
import sys; sys.path.insert(0, ".")
from flappie_amd import model as M
M.write_mdl("$d/flipflop5_r941native.h", M.synthetic_model(M.NET_LSTM5, 384, seed=1, ident="r941native"))
PY
export FLAPPIE_MODEL_DIR=$d FLAPPIE_CLI_TIMING=1
for mode in real emu emu_fast; do
  unset FFHIP_DEBUG_HOST_REHEARSAL_MSPS
  [ $mode = emu ] && export FFHIP_DEBUG_HOST_REHEARSAL_MSPS=104
  [ $mode = emu_fast ] && export FFHIP_DEBUG_HOST_REHEARSAL_MSPS=100000
  for n in 4096 16384; do
    echo "== $mode $n"
    bash -c "time flappie_amd/flappie --readers 12 --limit $n -o $d/out.fq $d/reads" 2>&1 | tail -22
  done
done
rm -rf $d
