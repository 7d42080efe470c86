#!/bin/bash
# Run ON THE GPU BOX: which rule turns FETCH_SIZE into bytes for the layer kernel's access patterns?
#  1. tools/variants/fetch_calib: 314.6 MB streamed once per launch in the kernel's own load forms -> FETCH_SIZE, TCC_EA0_RDREQ(_32B)
#  2. the H = 384 one-tile layer kernel as built, and with its sweep of h(t-1) compiled out (tools/variants/libffhip_nosweep.so =
#     tools/dev/experiments/lstm_split_ablation_switches.patch, -DFFHIP_SPLIT_ABLATE=8): if FETCH_SIZE does not move, h never comes over the fabric
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/traffic_rule; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- tools/variants/fetch_calib > $O/calib.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/calib_rdreq -- tools/variants/fetch_calib >> $O/calib.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- tools/variants/fetch_calib >> $O/calib.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/calib_wrreq -- tools/variants/fetch_calib >> $O/calib.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/layer_full -- python tools/dev/ablate.py full > $O/layer_full.log 2>&1
cp flappie_amd/libffhip.so /tmp/libffhip_keep.so
cp tools/variants/libffhip_nosweep.so flappie_amd/libffhip.so
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/layer_nosweep -- python tools/dev/ablate.py nosweep > $O/layer_nosweep.log 2>&1
cp /tmp/libffhip_keep.so flappie_amd/libffhip.so
python - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/traffic_rule"
print("# tools/dev/traffic_rule.sh -- counter values per launch (average over the launches of the run)")
print([l for l in open(O + "/calib.log").read().splitlines() if l.startswith("bytes per launch")][0])
for d in ("calib_fetch", "calib_rdreq", "calib_write", "calib_wrreq", "layer_full", "layer_nosweep"):
    acc = collections.defaultdict(list)
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if not (k.startswith("k_") or "k_lstm_split" in k or "ffhip" in k):
                continue
            if d.startswith("layer") and "k_lstm_split" not in k:
                continue
            acc[(k.split("(")[0][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-14s %-62s %-24s launches %3d  avg %.6g" % (d, k, c, len(v), sum(v) / len(v)))
for t in ("full", "nosweep"):
    print(open(O + "/layer_%s.log" % t).read().strip().splitlines()[-1])
PY
