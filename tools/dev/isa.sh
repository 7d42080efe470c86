#!/bin/bash
# usage: tools/dev/isa.sh FILE.hip KERNEL_SUBSTRING OUT.s [extra hipcc flags] -- device ISA of one kernel of a source file
f=$1; k=$2; out=$3; shift; shift; shift
d=$(mktemp -d); cd $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I/root/repo/flappie_amd/csrc -I/root/repo/include --save-temps "$@" -c /root/repo/flappie_amd/csrc/$f -o x.o 2>/dev/null
S=$(ls *gfx950*.s)
awk -v k="$k" '$0 ~ "^_Z.*"k".*:" && !f {f=1} f{print} f && /^\.Lfunc_end/{exit}' $S > $out
rm -rf $d; wc -l $out
