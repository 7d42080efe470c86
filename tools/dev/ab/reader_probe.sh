#!/bin/bash
# probe: host-only throughput of P flappie processes (GPU-free emulation at an unreachable rate: the readers are the limit)
N=16384
d=/dev/shm/rp; rm -rf $d; mkdir -p $d/one
for g in 0 1 2 3 4 5 6 7; do flappie_amd/fast5_tool synth $d/one $N 3500 5500 20260928 $g 8 > /dev/null & done; wait
python - <<PY
import sys; sys.path.insert(0, ".")
from flappie_amd import model as M
M.write_mdl("$d/flipflop5_r941native.h", M.synthetic_model(M.NET_LSTM5, 384, seed=1, ident="r941native"))
PY
export FLAPPIE_MODEL_DIR=$d FLAPPIE_CLI_TIMING=1 FFHIP_DEBUG_HOST_REHEARSAL_MSPS=100000 FFHIP_DEBUG_HOST_REHEARSAL_NOGPU=1
runp() {  # $1 = processes, $2 = label
  P=$1
  t0=$(python -c "import time; print(time.time())")
  for g in $(seq 0 $((P-1))); do
    flappie_amd/flappie --readers ${READERS:-12} --shard $g/8 $d/one -o $d/out.$g.fq > /dev/null 2> $d/err.$g &
  done; wait
  python - "$t0" "$P" "$2" "$d" <<'PY'
import sys, time
t0, P, label, d = float(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
wall = time.time() - t0
ph = {}
for ln in open(d + "/err.%d" % (P - 1)):
    p = ln.rsplit(None, 2)
    if len(p) == 3 and p[2] == "s":
        try: ph[p[0].strip()] = float(p[1])
        except ValueError: pass
print("%-58s P=%d wall %.2f s; last process: listed->done %.2f s = %.0f files/s per process, %.0f in all; fast5 read %.2f, waiting %.2f, list %.2f"
      % (label, P, wall, ph.get("files listed -> done", 0), 16384 / max(ph.get("files listed -> done", 1), 1e-9), P * 16384 / max(ph.get("files listed -> done", 1), 1e-9),
         ph.get("fast5 read", 0), ph.get("waiting for the reader", 0), ph.get("list files", 0)))
PY
}
for P in 1 2 4 8; do runp $P "default"; done
export MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=67108864
for P in 1 8; do runp $P "malloc: no mmap, no trim"; done
unset MALLOC_MMAP_THRESHOLD_ MALLOC_TRIM_THRESHOLD_ MALLOC_TOP_PAD_
for R in 2 6; do READERS=$R runp 8 "readers $R"; done
READERS=12 HDF5_USE_FILE_LOCKING=FALSE runp 8 "no HDF5 file locking"
nproc; grep -c processor /proc/cpuinfo; lscpu | grep -E "Model name|Socket|Thread|NUMA node\(s\)|Core" ; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | head -c 200
rm -rf $d
