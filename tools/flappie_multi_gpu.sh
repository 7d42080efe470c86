#!/bin/bash
# One flappie process per GPU over disjoint slices of the input files (the reference's README runs
# `find | parallel flappie` the same way on CPU cores, README.md:81-83).  No inter-process traffic.
#   tools/flappie_multi_gpu.sh NGPU OUT_PREFIX [flappie options ...] READS_DIR
# writes OUT_PREFIX.<gpu>.fq (and OUT_PREFIX.<gpu>.trace.hdf5 when FLAPPIE_TRACE=1).
set -euo pipefail
ngpu=$1; prefix=$2; shift 2
args=("$@")
reads=${args[-1]}
unset 'args[-1]'
here=$(cd "$(dirname "$0")/.." && pwd)
mapfile -t files < <(find "$reads" -name '*.fast5' | sort)
pids=()
for ((g = 0; g < ngpu; g++)); do
    slice=()
    for ((i = g; i < ${#files[@]}; i += ngpu)); do slice+=("${files[$i]}"); done
    [ ${#slice[@]} -eq 0 ] && continue
    extra=()
    [ "${FLAPPIE_TRACE:-0}" = 1 ] && extra=(--trace "$prefix.$g.trace.hdf5")
    FLAPPIE_HIP_DEVICE=$g "$here/flappie_amd/flappie" "${args[@]}" "${extra[@]}" -o "$prefix.$g.fq" "${slice[@]}" &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
