"""gaps between consecutive recurrent-layer launches in a rocprofv3 kernel trace (csv dir as argv[1])"""
import csv, glob, sys
import numpy as np
best = None
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
    if best is None or len(rows) > len(best): best = rows
rows = sorted(best)
L = [r for r in rows if "k_lstm_split" in r[2]]
gaps = np.array([(L[i + 1][0] - L[i][1]) / 1e3 for i in range(len(L) - 1)])
dur = np.array([(r[1] - r[0]) / 1e3 for r in L])
print("%d kernels, %d layer launches; layer duration mean %.1f us; gaps: mean %.1f us, median %.1f, p90 %.1f, max %.1f" % (len(rows), len(L), dur.mean(), gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max()))
print("queues of consecutive layer launches:", [r[3] for r in L[10:30]])
print("gaps (us) of launches 10..40:", [round(float(g), 1) for g in gaps[10:40]])
# what runs inside the gaps
i0 = L[12][1]
print("kernels between layer launch 12's end and launch 14's start:")
for r in rows:
    if r[0] >= i0 - 3000000 and r[0] <= L[14][0]:
        print("   %9.1f us  %8.1f us long  q %s  %s" % ((r[0] - i0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2]))
