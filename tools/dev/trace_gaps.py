"""busy / idle analysis of a rocprofv3 kernel trace (csv): union of kernel intervals, the largest idle gaps and what ran around them"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
last_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n if e >= cur_e else last_name
busy += cur_e - cur_s
print("kernels %d   span %.1f ms   busy (union) %.1f ms   idle %.1f ms in %d gaps" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))
by, cnt = {}, {}
for s, e, n in rows:
    by[n] = by.get(n, 0) + (e - s); cnt[n] = cnt.get(n, 0) + 1
for n, t in sorted(by.items(), key=lambda x: -x[1])[:14]:
    print("   %-52s %9.2f ms  %6d calls  avg %8.1f us" % (n, t / 1e6, cnt[n], t / cnt[n] / 1e3))
mid = {}
for g in gaps:
    if 2e4 <= g[0] < 2e6:
        k = (g[2], g[3]); mid[k] = (mid.get(k, (0, 0))[0] + g[0], mid.get(k, (0, 0))[1] + 1)
print("gaps of 20 us .. 2 ms by (kernel before -> kernel after):")
for k, (t, c) in sorted(mid.items(), key=lambda x: -x[1][0])[:12]:
    print("   %8.2f ms in %5d gaps   %s -> %s" % (t / 1e6, c, k[0], k[1]))
hist = [0, 0, 0, 0]
for g in gaps:
    hist[0 if g[0] < 2e4 else 1 if g[0] < 2e5 else 2 if g[0] < 2e6 else 3] += g[0]
print("idle by gap size: <20us %.1f ms, 20-200us %.1f ms, 0.2-2ms %.1f ms, >2ms %.1f ms" % tuple(h / 1e6 for h in hist))
for g in sorted(gaps, reverse=True)[:15]:
    print("   gap %8.2f ms at %8.1f ms   after %-40s before %s" % (g[0] / 1e6, g[1] / 1e6, g[2], g[3]))
