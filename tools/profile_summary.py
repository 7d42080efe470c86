#!/usr/bin/env python3
"""Turn rocprofv3 CSV outputs (kernel trace + separate FETCH_SIZE / WRITE_SIZE PMC passes) into the
summaries committed under profiles/.  usage: profile_summary.py TAG trace.csv fetch_counters.csv write_counters.csv"""
import collections
import csv
import json
import sys


def main():
    tag, trace, fetch, write = sys.argv[1:5]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in d.values())
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,percent"]
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        lines.append("%s,%d,%.3f,%.3f,%.3f,%.3f,%.2f" % (k, len(v), sum(v) / 1e3, sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    open("profiles/%s_kernel_stats.csv" % tag, "w").write(
        "# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline\n" + "\n".join(lines) + "\n")
    print("\n".join(lines[:10]))

    def pmc(path, name):
        out = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                out[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        return out
    f, w = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
    out = ["kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_read_MB_corrected(2xFETCH),hbm_write_MB"]
    res = {}
    for k in f:
        fa = sum(f[k]) / len(f[k])
        wa = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
        out.append("%s,%d,%.1f,%.1f,%.2f,%.2f" % (k, len(f[k]), fa, wa, 2 * fa * 1024 / 1e6, wa * 1024 / 1e6))
        res[k] = (2 * fa * 1024, wa * 1024)
    open("profiles/%s_hbm_traffic_pmc.csv" % tag, "w").write(
        "# separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --steps 4 --warmup 1\n"
        "# FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM)\n"
        + "\n".join(out) + "\n")
    print("\n".join(out))
    return res


if __name__ == "__main__":
    res = main()
    dom = [k for k in res if "k_lstm_split" in k] or [k for k in res if "k_lstm_fused" in k] or [k for k in res if "k_rnn_persist" in k]
    if dom:
        k = dom[0]
        H, nread, T, Tb = 384, 256, 4000, 800
        rnn_path = 3 if "split" in k else (2 if "fused" in k else 1)
        # algorithmic bytes: split kernel reads x and writes h at 6 B per value (three bf16 slices); fused f32 kernel 4 + 4;
        # unfused: read Xa (4H floats) + write h
        alg = Tb * nread * H * {3: 12, 2: 8, 1: 20}[rnn_path]
        entry = {"hidden": H, "nread": nread, "nsample": T, "rnn_path": rnn_path, "fused": rnn_path >= 2, "kernel": k,
                 "recurrent_layer_hbm_bytes_per_launch": int(res[k][0] + res[k][1]),
                 "read_bytes_corrected": int(res[k][0]), "write_bytes": int(res[k][1]),
                 "algorithmic_bytes_per_launch": alg, "source": "profiles/%s_hbm_traffic_pmc.csv" % sys.argv[1]}
        try:
            old = json.load(open("profiles/r01_traffic.json"))
            old = old if isinstance(old, list) else [old]
        except (OSError, ValueError):
            old = []
        old = [e for e in old if e.get("rnn_path", 2 if e.get("fused") else 1) != rnn_path] + [entry]
        json.dump(old, open("profiles/r01_traffic.json", "w"), indent=1)
        print(json.dumps(entry, indent=1))
