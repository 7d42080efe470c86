#!/usr/bin/env python3
"""From a rocprofv3 kernel trace CSV of bench.py: the time line around each recurrent stack (what runs between the last layer of
one batch and the first layer of the next)."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ffhip::", "").replace("ffhip::", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
t0 = rows[0][0]
rnn = [i for i, r in enumerate(rows) if "k_lstm_split" in r[2]]
# print the window between the 10th and 11th..15th rnn launches
lo = rnn[9] if len(rnn) > 16 else 0
hi = rnn[15] if len(rnn) > 16 else len(rows) - 1
for s, e, n, q in rows[lo:hi + 1]:
    print("%10.1f %10.1f %8.1f us  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
