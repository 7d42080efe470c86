#!/bin/bash
# Run ON THE GPU BOX: counter passes on the NON-layer kernels of a c2 step (k_conv_split, k_conv_small, k_head, CRF / decode chains), one batch
# at a time and unpaired so that each kernel has the chip to itself ("no SQ counter pass on this kernel exists").
# usage: tools/dev/pmc_front.sh TAG [bench flags]      -> gpurun_out/TAG_front_pmc.txt
tag=${1:-r04}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/pmc_front_$tag; rm -rf $O; mkdir -p $O
S="--config c2 --steps 6 --warmup 2 --inflight 1 --no-pair --no-cpu-baseline --no-h2d-leg --no-host-fed-leg $*"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python bench.py $S > $O/p$i.log 2>&1 || echo "pass $i ($set) failed" >> $O/failed.txt
done <<'SETS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
FETCH_SIZE
WRITE_SIZE
SETS
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py $S > $O/trace.log 2>&1
python - "$O" <<'PY' > gpurun_out/${tag}_front_pmc.txt
import csv, glob, collections, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "ffhip" not in k or "k_lstm" in k: continue
        k = k.split("ffhip::")[-1].split("(")[0]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "ffhip" not in k: continue
        dur[k.split("ffhip::")[-1].split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print("# tools/dev/pmc_front.sh: per-launch averages, c2 shape (256 reads x 4000 samples), one batch at a time, unpaired; us = rocprofv3 kernel-trace duration")
try: print("# failed passes:", open(O + "/failed.txt").read().strip().replace("\n", "; "))
except OSError: pass
for k in sorted(acc, key=lambda k: -sum(dur.get(k, [0])) / max(1, len(dur.get(k, [0])))):
    d = dur.get(k, [0.0])
    print("%s   launches %d   avg %.1f us" % (k, len(d), sum(d) / len(d)))
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    for n in sorted(c): print("    %-40s %.6g" % (n, c[n]))
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if n in c: print("    %-40s %.3f of wave cycles" % (n + " / SQ_WAVE_CYCLES", c[n] / wc))
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        print("    mfma busy fraction %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
cat gpurun_out/${tag}_front_pmc.txt
