#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for c in c2 c5 c4; do
timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_e_$c.json 2> gpurun_out/r02_e_$c.err
python - $c <<'PY'
import json, sys
d = json.load(open("gpurun_out/r02_e_%s.json" % sys.argv[1])); print(sys.argv[1], d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_e -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-h2d-leg > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/prof_r02_e/**/*kernel_trace.csv", recursive=True))[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:9]:
    print("%-50s %3d calls  avg %9.1f us" % (k[:50], len(v), sum(v) / len(v)))
PY
