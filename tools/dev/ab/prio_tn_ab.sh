#!/bin/bash
# decode-side priority (base = 2, variant dprio0 = 0) x FFHIP_DEBUG=conv1_tn=N at c4; and c2 / h256 / rle with both libraries
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp flappie_amd/libffhip.so /tmp/libffhip_base.so
one() { env $2 python bench.py --config $1 --steps $3 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['exposed_ms'], d['kernel_ms_per_step']['conv'])"; }
for rep in 1 2; do
for v in base dprio0; do
  if [ "$v" = base ]; then cp /tmp/libffhip_base.so flappie_amd/libffhip.so; else cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so; fi
  for tn in 4 2 1; do echo "$v c4 TN=$tn: $(one c4 FFHIP_DEBUG=conv1_tn=$tn 30)"; done
  echo "$v c2: $(one c2 X=1 120)"
  echo "$v h256: $(one h256 X=1 40)"
  echo "$v rle: $(one rle X=1 80)"
done
done
cp /tmp/libffhip_base.so flappie_amd/libffhip.so
