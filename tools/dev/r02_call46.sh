#!/bin/bash
cd $GRAFT_REPO_ROOT/flappie_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result"
cp ../libffhip.so /tmp/libffhip_orig.so
/opt/rocm/bin/hipcc $FL -DFFHIP_COUNT_FALLBACK -c ffhip_rnn_split.hip -o /tmp/split_v.o 2>/dev/null || echo build failed
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libffhip.so ffhip_kernels.o ffhip_rnn_persist.o /tmp/split_v.o ffhip_engine.o ffhip_layers.o ffhip_prep.o ffhip_rle.o
cd $GRAFT_REPO_ROOT
FFHIP_SKEW=1 timeout 200 python tools/dev/skew_diag.py
FFHIP_SKEW=0 timeout 200 python tools/dev/skew_diag.py
cp /tmp/libffhip_orig.so flappie_amd/libffhip.so
