#!/usr/bin/env python3
"""Per-launch averages of rocprofv3 --pmc counters for the recurrent layer kernel.  usage: pmc_layer.py DIR [DIR ...] (directories of
`rocprofv3 --kernel-trace --pmc ... --output-format csv -d DIR`); prints one line per counter."""
import collections, csv, glob, os, sys
for d in sys.argv[1:]:
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if "k_lstm_split" in r["Kernel_Name"] or "k_rnn_split" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0].replace("void ffhip::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print("%-40s %-34s launches %3d  avg %.6g" % (k, c, len(v), sum(v) / len(v)))
