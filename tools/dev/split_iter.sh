#!/bin/bash
# dev loop for ffhip_rnn_split.hip: build the library and the timeline harness, run both on a GPU box
set -e
cd /root/repo
make -C flappie_amd/csrc 2>&1 | grep -E "error|libffhip" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFFHIP_TIMELINE -Iflappie_amd/csrc flappie_amd/csrc/ffhip_rnn_split.hip tools/dev/split_timeline_main.cpp -o tools/bin/split_timeline 2>&1 | grep -E "error" || true
tag=${1:-x}
/usr/local/graft/bin/gpurun --timeout 900 -- "timeout 120 tools/bin/split_timeline 16 > gpurun_out/tl_$tag.log 2>&1; timeout 600 python tools/split_check.py ${2:-} > gpurun_out/split_$tag.log 2>&1; echo rc=\$?; head -32 gpurun_out/tl_$tag.log | cut -c 1-150; cat gpurun_out/split_$tag.log" 2>&1 | tail -48
