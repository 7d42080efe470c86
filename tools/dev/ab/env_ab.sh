#!/bin/bash
# usage: tools/dev/ab/env_ab.sh "ENV=1" config steps [config steps ...]  -- bench.py with and without an environment switch, interleaved twice per config
cd ${GRAFT_REPO_ROOT:-/root/repo}
sw=$1; shift
one() { env $1 python bench.py --config $2 --steps $3 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['exposed_ms'], d['kernel_ms_per_step']['viterbi_assembly'])"; }
while [ $# -ge 2 ]; do
  for rep in 1 2; do
    echo "$1 default: $(one X=1 $1 $2)"
    echo "$1 $sw: $(one $sw $1 $2)"
  done
  shift 2
done
