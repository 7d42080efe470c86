#!/bin/bash
# usage: tools/dev/regs.sh FILE.hip [grep pattern] [extra hipcc flags] -- VGPRs / spills / LDS of the kernels of one source file
f=$1; pat=${2:-.}; shift; shift
cd /root/repo/flappie_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/regs_test.o 2>&1 \
 | grep -E "error|Function Name|    VGPRs:|VGPR Spill|ScratchSize|LDS Size" | sed -E 's/^[^ ]+ (remark: )?[^ ]+ +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(l)print l; l=$0; next}{l=l" | "$0}END{print l}' | tr "\n" " " | sed "s/ | Name:/\nName:/g" | grep -E "$pat"
