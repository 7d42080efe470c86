#!/bin/bash
# One flappie process per GPU over disjoint slices of the input files (the reference's README runs
# `find | parallel flappie` the same way on CPU cores, README.md:81-83).  No inter-process traffic: reads are independent units.
#   tools/flappie_multi_gpu.sh NGPU OUT_PREFIX [flappie options ...] READS_DIR
# writes OUT_PREFIX.<slice>.fq (and OUT_PREFIX.<slice>.trace.hdf5 when FLAPPIE_TRACE=1).  Slice g takes files g, g + NGPU, ... of the
# sorted file list (flappie --shard g/NGPU), so `cat` of the slices is a permutation of the single-process output, each slice in input order.
# Every process binds itself and its reader children to the CPUs of ITS GPU's NUMA node before it forks them (flappie_cli.c: bind_to_gpu_numa, from
# FLAPPIE_HIP_DEVICE and /sys/class/drm/renderD*/device/numa_node; FLAPPIE_DEBUG=no_numa_bind leaves the placement to the scheduler).
# Environment:  FLAPPIE_DEVICES=0,1,...  device of each slice (default: slice g -> GPU g)
#               FLAPPIE_SERIAL=1         run the slices one after the other (several slices on ONE GPU: the persistent recurrent
#                                        kernels of two processes cannot share a GPU)
set -euo pipefail
ngpu=$1; prefix=$2; shift 2
args=("$@")
reads=${args[-1]}
unset 'args[-1]'
here=$(cd "$(dirname "$0")/.." && pwd)
[ -d "$reads" ] || { echo "$reads is not a directory" >&2; exit 2; }
IFS=',' read -r -a devs <<< "${FLAPPIE_DEVICES:-}"
pids=()
rc=0
for ((g = 0; g < ngpu; g++)); do
    slice=(--shard "$g/$ngpu" "$reads")          # the binary cuts its own slice from the sorted listing: no argument-length limit
    extra=()
    [ "${FLAPPIE_TRACE:-0}" = 1 ] && extra=(--trace "$prefix.$g.trace.hdf5")
    dev=${devs[$g]:-$g}
    if [ "${FLAPPIE_SERIAL:-0}" = 1 ]; then
        FLAPPIE_HIP_DEVICE=$dev "$here/flappie_amd/flappie" "${args[@]}" "${extra[@]}" -o "$prefix.$g.fq" "${slice[@]}" || rc=$?
    else
        FLAPPIE_HIP_DEVICE=$dev "$here/flappie_amd/flappie" "${args[@]}" "${extra[@]}" -o "$prefix.$g.fq" "${slice[@]}" &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
