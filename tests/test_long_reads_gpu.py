"""BASELINE.json config 5 against the oracle AT ITS REAL LENGTH: 100 000-sample reads -- 20 000 dependent steps per layer
at the r103 shape (LSTM5, H = 512), 50 000 blocks and 10 states at the r941_5mC shape (GRUmod5, H = 256).  The oracle needs ~165 s per such read
per core, so its answers are a committed fixture: tests/golden/long_reads.npz = tools/parity_long.py golden <- tools/parity_long.py oracle
(labelled oracle output, dot mode 0; calls, qualities, Viterbi paths and traces whole, transition scores and log posteriors at every 16th block).
The 16-read campaign over every block: profiles/r05_parity_c5_full.txt.  Reference path: networks.c:539-586, decode.c:119-204, 377-543."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
FIXTURE = os.path.join(ROOT, "tests", "golden", "long_reads.npz")


def test_fixture_is_oracle_output_prefix():
    """the committed fixture is what the oracle gives: the first 600 samples' worth of read 0 (a read's first blocks do not depend on its later samples
    in a FORWARD layer only, so this compares the generator's signal and the fixture's bookkeeping, not the network)"""
    import parity_long as PL
    recs = PL.load(FIXTURE)
    assert sorted(recs) == [("gru", 0)] + [("lstm", r) for r in range(4)]
    for (tag, r), rec in recs.items():
        nblk = (PL.T + (5 if tag == "lstm" else 2) - 1) // (5 if tag == "lstm" else 2)
        assert rec["sub"] == 16 and rec["path"].shape == (nblk + 1,) and rec["trans"].shape[0] == (nblk + 15) // 16
        assert len(str(rec["basecall"])) == len(str(rec["quality"])) > 15000
        assert PL.signal(tag, r).shape == (PL.T,)


@pytest.mark.gpu
def test_long_reads_against_the_oracle():
    """four r103-shape reads and one r941_5mC-shape read of 100 000 samples, default path, trace on: |dtrans| <= 1e-4 (north_star) wherever the fixture has a
    block -- printed with where along the read the worst one sits --, log posteriors within the suite's bound, Viterbi path / called bases / qualities equal
    (a read whose PATH differs is reported as a near-tie and must still agree on >= 99.9 % of its blocks), trace within one count with the rate printed."""
    import parity_long as PL
    from conftest import note_parity
    lines, tallies = PL.run_engine(PL.load(FIXTURE))
    for ln in lines:
        print(ln)
    for tag, t in tallies.items():
        note_parity(t["worst_trans"])          # (log posteriors: asserted below, not fed to the suite's summary -- over 20 000 / 50 000 blocks the ORACLE's own fp32
                                              # log-space sums carry more than the summary's 1e-4, decode.c:377-497)
        assert t["rnn_path"] == 3, "the split layer kernels are the path under test"
        assert t["worst_trans"] <= 1e-4, (tag, t)
        assert t["worst_post"] <= 2e-4, (tag, t)
        assert t["trace_more"] == 0 and t["trace_off1"] <= 2e-4 * t["trace_cells"], (tag, t)
        assert t["base_mismatch"] + t["path_mismatch"] + t["qual_mismatch"] == 0, (tag, t, lines)
