#!/usr/bin/env python3
"""Turn the rocprofv3 CSV outputs of tools/profile_config.sh (kernel trace, separate FETCH_SIZE / WRITE_SIZE passes, SQ counter
passes) into the summaries committed under profiles/.   usage: profile_summary.py TAG CONFIG SCRATCH_DIR OUT_DIR"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RECURRENT = ("k_lstm_split", "k_grumod_pack", "k_lstm_pack", "k_rnn_split", "k_lstm_fused", "k_rnn_persist")


def find(scratch, tag, leg, suffix):
    hits = sorted(glob.glob(os.path.join(scratch, "prof_%s_%s" % (tag, leg), "**", "*" + suffix), recursive=True))
    return max(hits, key=os.path.getsize) if hits else None          # (a traced command may have children: the bench process's file is the big one)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("ffhip::", "")


def main():
    tag, cfgname, scratch, outdir = sys.argv[1:5]
    import bench
    cfg = bench.CONFIGS[cfgname]
    cmd = "python bench.py --config %s --steps %s --warmup 2 --inflight 1 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg" % (cfgname, os.environ.get("PROFILE_STEPS", "4"))
    trace = find(scratch, tag, "trace", "kernel_trace.csv")
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in d.values())
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,percent"]
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        lines.append("\"%s\",%d,%.3f,%.3f,%.3f,%.3f,%.2f" % (k, len(v), sum(v) / 1e3, sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    open(os.path.join(outdir, "%s_kernel_stats.csv" % tag), "w").write(
        "# rocprofv3 --kernel-trace --stats --output-format csv -- %s\n" % cmd + "\n".join(lines) + "\n")
    print("\n".join(lines[:12]))
    rec = [k for k in d if any(n in k for n in RECURRENT)]
    dom = max(rec, key=lambda k: sum(d[k])) if rec else None

    def pmc(leg, names):
        path = find(scratch, tag, leg, "counter_collection.csv")
        out = {n: collections.defaultdict(list) for n in names}
        meta = {}
        if not path:
            return out, meta
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] in out:
                k = short(r["Kernel_Name"])
                out[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
                meta[k] = {c: r.get(c) for c in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if c in r}
        return out, meta

    f = pmc("fetch", ["FETCH_SIZE"])[0]["FETCH_SIZE"]
    w = pmc("write", ["WRITE_SIZE"])[0]["WRITE_SIZE"]
    out = ["kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_read_MB_corrected(2xFETCH),hbm_write_MB"]
    res = {}
    for k in f:
        fa = sum(f[k]) / len(f[k])
        wa = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
        out.append("\"%s\",%d,%.1f,%.1f,%.2f,%.2f" % (k, len(f[k]), fa, wa, 2 * fa * 1024 / 1e6, wa * 1024 / 1e6))
        res[k] = (2 * fa * 1024, wa * 1024, fa * 1024)
    open(os.path.join(outdir, "%s_hbm_traffic_pmc.csv" % tag), "w").write(
        "# separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE -- %s\n"
        "# FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM)\n" % cmd
        + "\n".join(out) + "\n")
    print("\n".join(out[:14]))

    # SQ counters of the recurrent kernels
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_MFMA", "SQ_INSTS_VALU",
             "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"]
    vals, meta = {}, {}
    for leg, ns in (("sq1", names[0:3]), ("sq2", names[3:6]), ("sq3", names[6:9])):
        o, m = pmc(leg, ns)
        vals.update(o)
        meta.update(m)
    sq = ["kernel,launches,avg_us," + ",".join(names) + ",mfma_busy_frac,grid,workgroup,lds,vgpr,agpr,sgpr"]
    for k in rec:
        row = []
        for n in names:
            v = vals.get(n, {}).get(k, [])
            row.append(sum(v) / len(v) if v else float("nan"))
        busy, gui = row[0], row[2]
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's SIMDs (x4 per CU on gfx950's counter: cycles with the matrix pipe busy);
        # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles -> busy fraction = busy / (GUI/8 * 1024 SIMDs)
        frac = busy / (gui / 8.0 * 1024.0) if gui == gui and gui > 0 else float("nan")
        m = meta.get(k, {})
        sq.append("\"%s\",%d,%.3f,%s,%.4f,%s" % (k, len(d[k]), sum(d[k]) / len(d[k]), ",".join("%.0f" % x for x in row), frac,
                                                  ",".join(str(m.get(c, "")) for c in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"))))
    open(os.path.join(outdir, "%s_sq_pmc.csv" % tag), "w").write(
        "# three separate passes (counter slots): rocprofv3 --kernel-trace --pmc {SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES} -- %s\n"
        "# per-launch averages; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)\n" % cmd + "\n".join(sq) + "\n")
    print("\n".join(sq))

    if dom and dom in res:
        H, nread, T = cfg["hidden"], cfg["nread"], cfg["nsample"]
        from flappie_amd import model as M
        Tb = M.synthetic_model(cfg["kind"], 128, seed=1).nblock(T)
        rnn_path = 3 if ("k_lstm_split" in dom or "k_grumod_pack" in dom or "k_lstm_pack" in dom) else (4 if "k_rnn_split" in dom else (2 if "fused" in dom else 1))
        G = 3 if cfg["kind"] == 1 else 4
        # algorithmic bytes per launch: split layer kernel reads x and writes h at 4 B per value (two fp16 slices); f32 fused 4 + 4;
        # recurrence-only kernels read the projected gates (G*H floats) and write h
        paired = 2 if "_pair" in dom else 1                 # k_lstm_split_pair: one launch carries the layer of two batches
        alg = Tb * nread * H * {3: 8, 2: 8, 1: 4 * G + 4, 4: 4 * G + 4}[rnn_path] * paired
        # Round 4 settled the rule with a microbenchmark in the kernel's own load forms (tools/dev/fetch_calib.cpp, profiles/r04_traffic_rule.txt):
        # 314.6 MB streamed once through global_load_dwordx4, buffer_load_dwordx4 sc1, buffer_load ... lds sc1 and one dword per line reads as
        # FETCH_SIZE 157.3 MB with TCC_EA0_RDREQ = one request per 128-byte LINE -- FETCH_SIZE is HALF the bytes for every read form of this
        # kernel (MI355X_MICROARCH.md's rule; round 3's "raw" reading was wrong: its 4.92 M "sectors" were 2.46 M lines of x plus as many of h);
        # WRITE_SIZE is the byte count (64-byte requests).  Traffic = 2 x FETCH_SIZE + WRITE_SIZE.
        rd = res[dom][0]
        entry = {"config": cfgname, "kind": cfg["kind"], "hidden": H, "nread": nread, "nsample": T, "rnn_path": rnn_path, "fused": rnn_path in (2, 3), "kernel": dom,
                 "reads_per_launch": nread * paired,
                 "recurrent_layer_hbm_bytes_per_launch": int(rd + res[dom][1]),
                 "read_bytes": int(rd), "write_bytes": int(res[dom][1]),
                 "read_bytes_if_fetch_size_doubled": int(res[dom][0]), "hbm_bytes_if_fetch_size_doubled": int(res[dom][0] + res[dom][1]),
                 "fetch_size_rule": "2 x FETCH_SIZE + WRITE_SIZE (calibrated in this kernel's own load forms: profiles/r04_traffic_rule.txt)",
                 "algorithmic_bytes_per_launch": alg, "source": "profiles/%s_hbm_traffic_pmc.csv" % tag}
        json.dump([entry], open(os.path.join(outdir, "%s_traffic.json" % tag), "w"), indent=1)
        print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
