#!/usr/bin/env python3
"""The results table of a round, generated from the tracked records under profiles/ (no figure in the documents that the records do not hold).

    tools/make_tables.py [r06]            prints the table
    tools/make_tables.py [r06] --write    and puts it between the markers <!-- results:r06 --> ... <!-- /results:r06 --> of DESIGN.md, README.md and BASELINE.md

tests/test_docs_tables.py fails when a document's block differs from what the records give.  Every cell names where it comes from in the header row's legend:
  value, ms per step, exposed ms, roofline.frac, layer launch (HIP events), cpu_baseline     profiles/TAG_<config>_bench.json   (`python bench.py --config <config> --no-host-fed-leg`)
  layer launch (rocprofv3)                                                                   profiles/TAG_<config>_kernel_stats.csv (the dominant recurrent kernel's avg_us)
  matrix pipe busy, VALU instructions                                                        profiles/TAG_<config>_sq_pmc.csv
  HBM traffic per layer launch, measured / algorithmic                                       profiles/TAG_<config>_traffic.json
  the driver's command                                                                       profiles/TAG_bench_default.json"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "README.md", "BASELINE.md")
CONFIGS = ("c2", "h256", "c4", "c5", "rle")
RECURRENT = ("k_lstm_split", "k_grumod_pack", "k_lstm_pack", "k_rnn_split", "k_lstm_fused", "k_rnn_persist")


def _rows(path):
    with open(path) as fh:
        return list(csv.DictReader(ln for ln in fh if not ln.startswith("#")))


def table(tag):
    P = lambda name: os.path.join(ROOT, "profiles", name)      # noqa: E731
    out = ["| config | Msamples/s | ms per step | layer launch, HIP events (rocprofv3) | `roofline.frac` | matrix pipe busy | VALU instructions per launch | exposed ms | HBM per launch: measured / algorithmic | `cpu_baseline` (16 CPUs) |",
           "|---|---|---|---|---|---|---|---|---|---|"]
    for c in CONFIGS:
        b = json.load(open(P("%s_%s_bench.json" % (tag, c))))
        r = b["roofline"]
        ks = [x for x in _rows(P("%s_%s_kernel_stats.csv" % (tag, c))) if any(n in x["kernel"] for n in RECURRENT)]
        dom = max(ks, key=lambda x: float(x["total_ms"]))
        sq = [x for x in _rows(P("%s_%s_sq_pmc.csv" % (tag, c))) if x["kernel"] == dom["kernel"]]
        tr = json.load(open(P("%s_%s_traffic.json" % (tag, c))))[0]
        out.append("| `%s` | **%.1f** | %.2f | %.3f ms (%.3f) `%s` | %.3f | %s | %s | %.2f | %.2f / %.2f GB | %.3f |"
                   % (c, b["value"], b["ms_per_step"], r["avg_launch_ms"], float(dom["avg_us"]) / 1e3, dom["kernel"].split("<")[0], r["frac"],
                      ("%.3f" % float(sq[0]["mfma_busy_frac"])) if sq else "-", ("%.0f M" % (float(sq[0]["SQ_INSTS_VALU"]) / 1e6)) if sq else "-",
                      b["exposed_ms"], tr["recurrent_layer_hbm_bytes_per_launch"] / 1e9, tr["algorithmic_bytes_per_launch"] / 1e9, b["cpu_baseline"]["value"]))
    d = json.load(open(P("%s_bench_default.json" % tag)))
    hf = d.get("host_fed") or {}
    lm = d.get("length_mix") or {}
    with open(P("%s_bench_default_kernel_stats.csv" % tag)) as fh:
        dk = [x for x in csv.DictReader(ln for ln in fh if not ln.startswith("#")) if any(n in x["Name"] for n in RECURRENT)]
    dk = max(dk, key=lambda x: float(x["TotalDurationNs"]))
    out.append("")
    out.append("The driver's command (`python bench.py`, `profiles/%s_bench_default.json`): **%.1f Msamples/s**, %.3f ms per step, layer launch %.3f ms by HIP events (rocprofv3 `--stats` of the same command, `%s_bench_default_kernel_stats.csv`: %.3f ms over %s calls), `roofline.frac` %.4f, "
               "`exposed_ms` %.3f, `decode_hbm` %.0f GB/s, `h2d_inclusive` %.1f, `host_fed` %s Msamples/s (the `flappie` binary from fast5 files of 3500-5500 samples), `length_mix` %s (the same binary on log-normal read lengths, 1000 … 200 000 samples: its steady state by its own time stamps, its padding efficiency), `cpu_baseline` %.3f Msamples/s on %d CPUs (`%s`)."
               % (tag, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], tag, float(dk["AverageNs"]) / 1e6, dk["Calls"], d["roofline"]["frac"], d["exposed_ms"], d["decode_hbm"]["achieved"],
                  (d.get("h2d_inclusive") or {}).get("value", float("nan")), ("%.1f" % hf["value"]) if hf.get("value") else "-",
                  ("%.1f Msamples/s at %.2f" % (lm["value"], lm["padding_efficiency"])) if lm.get("value") else "-", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"]))
    out.append("Records: `profiles/%s_{c2,h256,c4,c5,rle}_{bench.json, kernel_stats.csv, sq_pmc.csv, hbm_traffic_pmc.csv, traffic.json}` (`tools/profile_all.sh`, one box, the final tree); "
               "the boxes of the pool differ by a few per cent (the layer launches run at the socket's power cap), so do other runs of these commands." % tag)
    return "\n".join(out)


def block_re(tag):
    return re.compile(r"(<!-- results:%s -->\n)(.*?)(<!-- /results:%s -->)" % (tag, tag), re.S)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r06"
    t = table(tag)
    print(t)
    if "--write" in sys.argv:
        for doc in DOCS:
            p = os.path.join(ROOT, doc)
            s = open(p).read()
            if not block_re(tag).search(s):
                print("%s: no <!-- results:%s --> block" % (doc, tag), file=sys.stderr)
                continue
            open(p, "w").write(block_re(tag).sub(lambda m: m.group(1) + t + "\n" + m.group(3), s))


if __name__ == "__main__":
    main()
