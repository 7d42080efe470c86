#!/bin/bash
# same-box A/B of the front ordering (FFHIP_DEBUG=front_order=..., streams=N) and of two batches in flight for the full-chip shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() {   # label, env string, bench args
    local out; out=$(env $2 python bench.py --no-cpu-baseline --no-h2d-leg --no-host-fed-leg $3 2>/dev/null | tail -1)
    python - "$1" "$out" <<'PY'
import json,sys
d=json.loads(sys.argv[2]); print("%-34s %8.2f Msamples/s  %8.3f ms/step  exposed %.3f  layer %.3f ms"%(sys.argv[1],d['value'],d['ms_per_step'],d.get('exposed_ms',0),d['roofline']['avg_launch_ms']))
PY
}
for rep in 1 2; do
one "c2 batch-order, 2 streams" "FFHIP_DEBUG=front_order=batch,streams=2" "--config c2"
one "c2 layer-order, no decode wait" "FFHIP_DEBUG=no_decode_wait" "--config c2"
one "c2 layer-order, 4 streams" "FFHIP_DEBUG=streams=4" "--config c2"
one "rle layer-order, 4 streams" "FFHIP_DEBUG=streams=4" "--config rle"
one "rle batch-order, 2 streams" "FFHIP_DEBUG=front_order=batch,streams=2" "--config rle"
for c in h256 c4; do
one "$c one in flight" "FFHIP_DEBUG=streams=2" "--config $c --inflight 1 --steps 30"
one "$c two, batch-order" "FFHIP_DEBUG=front_order=batch,streams=2" "--config $c --inflight 2 --steps 30"
one "$c two, layer-order" "FFHIP_DEBUG=streams=4" "--config $c --inflight 2 --steps 30"
done
done
one "c5 one in flight" "FFHIP_DEBUG=streams=2" "--config c5 --inflight 1 --steps 6 --warmup 2"
one "c5 two, layer-order" "FFHIP_DEBUG=streams=4" "--config c5 --inflight 2 --steps 6 --warmup 2"
