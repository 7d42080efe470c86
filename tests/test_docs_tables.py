"""The results tables of DESIGN.md, README.md and BASELINE.md are GENERATED from the tracked records under profiles/ (tools/make_tables.py):
a figure in a document that the records do not hold fails here."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_results_blocks_are_what_the_records_give():
    import make_tables
    tag = "r06"
    want = make_tables.table(tag)
    for doc in make_tables.DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        m = make_tables.block_re(tag).search(text)
        assert m, "%s has no <!-- results:%s --> block" % (doc, tag)
        assert m.group(2).rstrip("\n") == want, "%s: the results block differs from the records (run tools/make_tables.py %s --write)" % (doc, tag)


def test_every_record_the_table_names_is_tracked():
    for c in ("c2", "h256", "c4", "c5", "rle"):
        for suffix in ("bench.json", "kernel_stats.csv", "sq_pmc.csv", "hbm_traffic_pmc.csv", "traffic.json"):
            assert os.path.exists(os.path.join(ROOT, "profiles", "r06_%s_%s" % (c, suffix))), (c, suffix)
    for name in ("r06_bench_default.json", "r06_bench_default_kernel_stats.csv", "r06_gates_c2.txt", "r06_gates_h256.txt", "r06_gates_c4.txt", "r06_gate_speed.txt",
                 "r06_length_mix.txt", "r06_pack_bench.txt"):
        assert os.path.exists(os.path.join(ROOT, "profiles", name)), name
