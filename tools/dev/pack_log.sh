d=/dev/shm/pl; rm -rf $d; mkdir -p $d/reads
flappie_amd/fast5_tool synthln $d/reads 49152 8000 1.0 1000 200000 20260930 > /dev/null
python -c "
import sys; sys.path.insert(0,'.')
from flappie_amd import model as M
M.write_mdl('$d/flipflop5_r941native.h', M.synthetic_model(M.NET_LSTM5, int('${PL_H:-384}'), seed=1, ident='r941native'))"
FLAPPIE_MODEL_DIR=$d FLAPPIE_DEBUG=pack_log FLAPPIE_CLI_TIMING=1 flappie_amd/flappie --readers 4 -o $d/out.fq $d/reads 2>&1 | grep "^packed batch\|^batches\|files listed"
rm -rf $d
