#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the flappie binary on a directory of log-normal read lengths (where does a packed chunk's time go?)
# usage: tools/dev/mixed_trace.sh [hidden=384] [nfiles=32768]
H=${1:-384}; N=${2:-32768}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
D=/dev/shm/lmtrace; rm -rf $D; mkdir -p $D/mixed
python -c "
import sys; sys.path.insert(0,'.')
from flappie_amd import model as M
M.write_mdl('$D/flipflop5_r941native.h', M.synthetic_model(M.NET_LSTM5, $H, seed=1, ident='r941native'))"
flappie_amd/fast5_tool synthln $D/mixed $N 8000 1.0 1000 200000 20260929
export FLAPPIE_MODEL_DIR=$D FLAPPIE_CLI_TIMING=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mixed -- flappie_amd/flappie --readers 4 -o $D/out.fq $D/mixed 2> gpurun_out/mixed_trace_stderr.txt
f=$(find gpurun_out/prof_mixed -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -d, -f1-6
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_mixed/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
iv = sorted(zip(st, en)); busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels %d; first start -> last end %.3f s; union of kernel intervals %.3f s" % (len(rows), (max(en) - min(st)) / 1e9, busy / 1e9))
PY
tail -16 gpurun_out/mixed_trace_stderr.txt
rm -rf $D
