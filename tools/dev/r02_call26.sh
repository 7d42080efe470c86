#!/bin/bash
# does the corruption follow the reading LANES or the block ADDRESSES?  k_transpost8 with blk = tid ^ 32
cd $GRAFT_REPO_ROOT/flappie_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result"
cp ../libffhip.so /tmp/libffhip_orig.so
for V in "-DFFHIP_DBG_NOPK"; do
  /opt/rocm/bin/hipcc $FL $V -c ffhip_kernels.hip -o /tmp/kern_v.o 2>/dev/null || { echo "build failed $V"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libffhip.so /tmp/kern_v.o ffhip_rnn_persist.o ffhip_rnn_split.o ffhip_engine.o ffhip_layers.o ffhip_prep.o ffhip_rle.o
  echo "== variant $V"
  (cd $GRAFT_REPO_ROOT; QUICK=1 NREAD=256 timeout 300 python tools/dev/inflight_diag2.py 2>&1 | tail -1)
done
cp /tmp/libffhip_orig.so ../libffhip.so
