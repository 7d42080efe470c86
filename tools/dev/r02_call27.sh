#!/bin/bash
cd $GRAFT_REPO_ROOT/flappie_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result"
cp ../libffhip.so /tmp/libffhip_orig.so
for V in "-DFFHIP_DBG_NOMFMA" "-DFFHIP_DBG_NOMFMA -DFFHIP_SPLIT_ABLATE=4"; do
  /opt/rocm/bin/hipcc $FL $V -c ffhip_rnn_split.hip -o /tmp/split_v.o 2>/dev/null || { echo "build failed $V"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libffhip.so ffhip_kernels.o ffhip_rnn_persist.o /tmp/split_v.o ffhip_engine.o ffhip_layers.o ffhip_prep.o ffhip_rle.o
  echo "== variant $V"
  (cd $GRAFT_REPO_ROOT; QUICK=1 NREAD=256 timeout 300 python tools/dev/inflight_diag2.py 2>&1 | tail -1)
done
cp /tmp/libffhip_orig.so ../libffhip.so
