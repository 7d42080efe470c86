#!/bin/bash
# Run ON THE GPU BOX: the flappie binary on a generated directory of log-normal read lengths, phases and packed-batch shapes on stderr.   usage: tools/dev/mixed_run.sh [hidden=384] [nfiles=65536] [extra env ...]
H=${1:-384}; N=${2:-65536}; shift; shift
MODEL=""; KIND=NET_LSTM5; NAME=flipflop5_r941native.h; IDENT=r941native
if [ "$H" = g256 ]; then H=256; MODEL="--model r941_5mC"; KIND=NET_GRUMOD5; NAME=flipflop_r941native5mC.h; IDENT=r941native5mC; fi
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"
D=/dev/shm/lmrun; rm -rf $D; mkdir -p $D/mixed
python -c "
import sys; sys.path.insert(0,'.')
from flappie_amd import model as M
M.write_mdl('$D/$NAME', M.synthetic_model(M.$KIND, $H, seed=1, ident='$IDENT'))"
flappie_amd/fast5_tool synthln $D/mixed $N 8000 1.0 1000 200000 20260929
env "$@" FLAPPIE_MODEL_DIR=$D FLAPPIE_CLI_TIMING=1 flappie_amd/flappie $MODEL --readers 4 -o $D/out.fq $D/mixed
rm -rf $D
