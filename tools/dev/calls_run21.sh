# round 5, call 21: (a) gap timelines of every config at its default in-flight mode; (b) the host-fed leg's spread on one box
mkdir -p gpurun_out/r05n; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in c2 h256 c4 rle; do
  rm -rf /tmp/tr_$c
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$c -- python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg > gpurun_out/r05n/trace_$c.log 2>&1
  echo "== $c" ; python tools/dev/pair_timeline.py /tmp/tr_$c
done > gpurun_out/r05n/timelines.txt 2>&1
for rep in 1 2; do
  timeout 600 python bench.py --steps 60 --warmup 3 --no-cpu-baseline 2>gpurun_out/r05n/bench$rep.err | tail -1 > gpurun_out/r05n/bench$rep.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r05n/bench$rep.json")); print("bench $rep: value %.2f host_fed %s" % (d["value"], json.dumps(d.get("host_fed"))[:700]))
PY
done > gpurun_out/r05n/hostfed.txt 2>&1
tail -5 gpurun_out/r05n/hostfed.txt; head -60 gpurun_out/r05n/timelines.txt
