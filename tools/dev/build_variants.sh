#!/bin/bash
# build library variants of the split layer kernel for same-device A/B timing:
#   tools/dev/build_variants.sh tag1="-DX=1" tag2="-DX=2" ...   ->  tools/variants/libffhip_<tag>.so
# then on the GPU box: tools/dev/ab_run.sh tag1 tag2 ...  (copies each over flappie_amd/libffhip.so in turn, interleaved)
set -e
cd /root/repo
make -C flappie_amd/csrc 2>&1 | grep -E "error|warning: unused" || true
mkdir -p tools/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Iflappie_amd/csrc"
OTHERS=$(ls flappie_amd/csrc/*.o | grep -v ffhip_rnn_split.o)
for spec in "$@"; do
  tag=${spec%%=*}; defs=${spec#*=}
  /opt/rocm/bin/hipcc $FLAGS $defs -c flappie_amd/csrc/ffhip_rnn_split.hip -o tools/variants/split_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/libffhip_$tag.so $OTHERS tools/variants/split_$tag.o
  echo built tools/variants/libffhip_$tag.so
done
