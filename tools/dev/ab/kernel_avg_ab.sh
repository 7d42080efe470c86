#!/bin/bash
# usage: tools/dev/ab/kernel_avg_ab.sh "<kernel name substrings, |-separated>" base variant ...   -- the default bench under rocprofv3 --stats with each
# library in turn: average durations of the named kernels in the PIPELINE (beside whatever the other streams run), and the plain bench value
cd ${GRAFT_REPO_ROOT:-/root/repo}
pat=$1; shift
cp flappie_amd/libffhip.so /tmp/libffhip_base.so
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libffhip_base.so flappie_amd/libffhip.so; else cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so; fi
  val=$(python bench.py --config ${CFG:-c2} --steps ${STEPS:-120} --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['exposed_ms'])")
  rm -rf /tmp/kab; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kab -- python $OLDPWD/bench.py --config ${CFG:-c2} --steps 40 --warmup 4 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg > /dev/null 2>&1)
  avg=$(python - "$pat" <<'PY'
import csv,glob,sys
pats=sys.argv[1].split("|")
for f in glob.glob("/tmp/kab/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        for p in pats:
            if p in r["Name"]: print("%s %.1f us;"%(p,float(r["AverageNs"])/1e3),end=" ")
PY
)
  echo "$v: $val Msamples/s, exposed ms   | $avg"
done
done
cp /tmp/libffhip_base.so flappie_amd/libffhip.so
