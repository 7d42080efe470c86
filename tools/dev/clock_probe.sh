#!/bin/bash
# samples clocks / power / temperature while the default bench runs (is the layer kernel clock- or power-limited?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction)" | head -12
echo "---- under load"
python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-host-fed-leg --no-h2d-leg > /tmp/b.json 2>/dev/null &
pid=$!
sleep 9
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|junction" | tr '\n' ' ' | sed 's/=\+//g; s/  */ /g'; echo
  sleep 1
done
wait $pid
python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
rocm-smi --showperflevel --showpowerprofile 2>/dev/null | grep -v "^=" | head -12
