#!/bin/bash
# on the GPU box: interleaved timing of library variants (three rounds), then the base library is restored
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp flappie_amd/libffhip.so /tmp/libffhip_base.so
for round in 1 2 3; do
  for tag in "$@"; do
    cp tools/bin/libffhip_$tag.so flappie_amd/libffhip.so
    timeout 300 python tools/dev/layer_time.py $tag 2>&1 | tail -1
  done
done
cp /tmp/libffhip_base.so flappie_amd/libffhip.so
