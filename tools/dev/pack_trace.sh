#!/bin/bash
# kernel trace of the flappie binary on the length mix (run on the GPU box): what runs between the layer launches of two packed batches
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
d=/dev/shm/pl; rm -rf $d; mkdir -p $d/reads
flappie_amd/fast5_tool synthln $d/reads 40000 8000 1.0 1000 200000 20260930 > /dev/null
python -c "
import sys; sys.path.insert(0,'.')
from flappie_amd import model as M
M.write_mdl('$d/flipflop5_r941native.h', M.synthetic_model(M.NET_LSTM5, 384, seed=1, ident='r941native'))"
rm -rf gpurun_out/prof_packtrace
FLAPPIE_MODEL_DIR=$d rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_packtrace -- flappie_amd/flappie --readers 4 -o $d/out.fq $d/reads > /dev/null 2>&1
rm -rf $d
f=$(find gpurun_out/prof_packtrace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ffhip::", "")) for r in rows))
layers = [e for e in ev if e[2].startswith("k_lstm_split") or e[2].startswith("k_lstm_pack")]
long_l = [e for e in layers if e[1] - e[0] > 100e6]         # the layer launches of the full batches (> 100 ms)
print("%d kernels, %d layer launches, %d of them longer than 100 ms" % (len(ev), len(layers), len(long_l)))
# gaps between consecutive long layer launches that are longer than 20 ms = between two batches
for a, b in zip(long_l, long_l[1:]):
    gap = b[0] - a[1]
    if gap < 20e6: continue
    print("\n== gap of %.1f ms between layer launches (the launch before: %.1f ms)" % (gap / 1e6, (a[1] - a[0]) / 1e6))
    inside = [e for e in ev if e[0] >= a[1] - 1e6 and e[1] <= b[0] + 1e6 and e is not a and e is not b]
    agg = {}
    for s, e, n in inside:
        k = agg.setdefault(n, [0, 0.0, 1e18, 0]); k[0] += 1; k[1] += (e - s) / 1e6; k[2] = min(k[2], s); k[3] = max(k[3], e)
    for n, (c, t, s0, e1) in sorted(agg.items(), key=lambda kv: kv[1][2]):
        print("  %-60s x%-4d %8.2f ms in all   first start %+8.2f ms, last end %+8.2f ms (after the layer launch's end)" % (n[:60], c, t, (s0 - a[1]) / 1e6, (e1 - a[1]) / 1e6))
P
