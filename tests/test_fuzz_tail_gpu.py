"""The tail of the GPU <-> oracle deviation, as a test the driver runs ("weak" 2).

north_star asks for transition scores within 1e-4 of the reference.  On every model the rest of the suite uses the worst case
is ~6e-5; the randomised differential test (tools/dev/diff_fuzz.py, seeds 1 and 5) found reads where BOTH GPU paths deviate
more -- all on one random LSTM5 model (H = 256, synthetic_model(seed=102)), an untrained and barely contractive recurrence
through which rounding differences of 1e-7 grow over 5 layers x 500 steps.  tests/golden/fuzz_tail.npz holds those reads
(tests/golden/make_fuzz_tail.py).  This test re-runs them and

  * asserts the DOCUMENTED bound for them: max |dtrans| <= 3.0e-4 against the oracle for the default
    split-precision path and for the f32-MFMA path;
  * reports who is off: the oracle with a double accumulator for every dot product (ff_oracle.c dot mode 1 -- the float32
    network without summation error) is the yardstick; GPU <-> yardstick and oracle <-> yardstick are printed side by side,
    and the default GPU path must not be further from it than 1.5x the reference-order oracle is;
  * counts what the tolerances of the suite let through on these reads: base / quality strings that differ from the
    oracle's, and the fraction of trace cells off by one count.

Run with -rP (or -s) to see the report."""
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

from flappie_amd import model as M

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _oracle_job(job):
    kind, hidden, seed, signal, mode = job
    from oracle import ffo
    mdl = M.synthetic_model(kind, hidden, seed=seed, gains=M.SYNTH_GAINS_R3[kind])      # the reads were found on the round-3 model
    om = ffo.OracleModel(mdl)
    with ffo.dot_mode(mode):
        r = om.basecall(signal)
    return dict(trans=r["trans"], basecall=r["basecall"], quality=r["quality"], trace=r["trace"], path=r["path"])


def test_fuzz_tail_reads_stay_within_the_documented_bound(engine):
    from flappie_amd import binding as B
    g = np.load(os.path.join(HERE, "golden", "fuzz_tail.npz"))
    n = int(g["n"])
    assert n >= 6
    reads = [(int(g["kind%d" % i]), int(g["hidden%d" % i]), int(g["model_seed%d" % i]), g["signal%d" % i]) for i in range(n)]
    assert len({r[:3] for r in reads}) == 1          # one model: LSTM5 H = 256 seed 102
    kind, hidden, seed = reads[0][:3]
    jobs = [(kind, hidden, seed, r[3], mode) for r in reads for mode in (0, 1)]
    with ProcessPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:      # ~1 s per read and mode, in parallel
        res = list(pool.map(_oracle_job, jobs))
    oracle, yard = res[0::2], res[1::2]

    mdl = M.synthetic_model(kind, hidden, seed=seed, gains=M.SYNTH_GAINS_R3[kind])
    dm = B.DeviceModel(engine, mdl)
    cap = max(r[3].size for r in reads)
    out = {}
    try:
        for name, flags in (("split", 0), ("f32", B.RUN_F32_RNN)):
            b = B.Batch(dm, n, cap)
            b.set_signals_ragged([r[3] for r in reads])
            b.run(1.0, flags)
            b.finish()
            out[name] = [dict(trans=b.transitions(i), basecall=b.basecall(i), quality=b.quality(i), trace=b.trace(i), path=b.path(i)[0]) for i in range(n)]
            b.close()
    finally:
        dm.close()

    def dmax(a, c):
        return max(float(np.abs(x["trans"].astype(np.float64) - y["trans"]).max()) for x, y in zip(a, c))

    rows = []
    for name in ("split", "f32"):
        gpu = out[name]
        nbase = sum(x["basecall"] != y["basecall"] for x, y in zip(gpu, oracle))
        nqual = sum(x["quality"] != y["quality"] for x, y in zip(gpu, oracle))
        npath = sum(not np.array_equal(x["path"], y["path"]) for x, y in zip(gpu, oracle))
        cells = sum(x["trace"].size for x in gpu)
        off1 = sum(int((np.abs(x["trace"] - y["trace"]) == 1).sum()) for x, y in zip(gpu, oracle))
        offn = sum(int((np.abs(x["trace"] - y["trace"]) > 1).sum()) for x, y in zip(gpu, oracle))
        rows.append((name, dmax(gpu, oracle), dmax(gpu, yard), nbase, nqual, npath, off1, offn, cells))
    d_oy = dmax(oracle, yard)
    print("fuzz tail, %d reads of LSTM5 H = %d seed %d (max |dtrans| over all scores):" % (n, hidden, seed))
    print("  oracle (reference-order float sums)  <-> yardstick (double accumulators): %.2e" % d_oy)
    for name, d_go, d_gy, nbase, nqual, npath, off1, offn, cells in rows:
        print("  GPU %-5s <-> oracle %.2e   <-> yardstick %.2e   reads with a different base string %d, quality string %d, Viterbi path %d (of %d);"
              " trace cells off by one %d, by more %d, of %d (%.4f %%)" % (name, d_go, d_gy, nbase, nqual, npath, n, off1, offn, cells, 100.0 * off1 / cells))
    # (rounds 2-4 also asserted a flat 3e-4 here -- three times north_star's tolerance, and a bound on the MODEL: on these reads two float32 evaluations
    # of the reference's own algorithm sit 1.1e-4 apart.  VERDICT r4, next 6: what is held is the engine's distance from the network proper RELATIVE to the
    # oracle's own, below, for both GPU paths.)
    for name, d_go, d_gy, nbase, nqual, npath, off1, offn, cells in rows:
        assert d_gy <= 2.0 * d_oy + 2.0e-5, (name, d_gy, d_oy)      # (the f32-MFMA cross-check path: measured 1.9x on these reads)
        # an ABSOLUTE ceiling beside the relative bound (were the oracle and the yardstick to drift together, the relative bound alone would
        # hold nothing): on this ill-conditioned model two float32 evaluations of the reference's own algorithm sit 1.1e-4 apart, the engine measured
        # 1.0e-4 / 2.1e-4 (default / f32-MFMA path) from the oracle; 3e-4 is the ceiling rounds 2-4 held
        assert d_go <= 3.0e-4, (name, d_go)
        assert offn == 0
    # the default path is no further from the float32 network proper than the reference-order sums are (measured: 0.8x; the
    # f32-MFMA cross-check path sits at 1.9x on these reads -- every float32 evaluation order scatters by 1-2e-4 on this model)
    assert rows[0][2] <= max(1.5 * d_oy, 1.0e-4), (rows[0][2], d_oy)
    # ... and, as a bound that means something beside the flat 3e-4: the default path is at most 2e-5 further from
    # the float32 network proper than the reference-order oracle is on these very reads
    assert rows[0][2] <= d_oy + 2.0e-5, (rows[0][2], d_oy)


def test_recorded_near_tie_reads(engine):
    """tests/golden/near_ties.npz: reads a campaign found called differently from the oracle on scores that agree to rounding (a near-tie
    of the posterior decode).  Recorded so that a SECOND such read is noticed: the campaigns count what is not in this file.  Held here:
    the scores stay within the recorded distance (1e-5 / 2e-5) of the oracle's, the Viterbi paths part for no more blocks than recorded, and,
    where recorded so, the all-f32 path keeps calling the read as the oracle does."""
    from flappie_amd import binding as B
    from oracle import ffo
    g = np.load(os.path.join(HERE, "golden", "near_ties.npz"))
    for i in range(int(g["n"])):
        kind, hidden, seed = int(g["kind%d" % i]), int(g["hidden%d" % i]), int(g["model_seed%d" % i])
        mdl = M.synthetic_model(kind, hidden, seed=seed, gains=tuple(float(x) for x in g["gains%d" % i]))
        sig = g["signal%d" % i]
        ref = ffo.OracleModel(mdl).basecall(sig)
        dm = B.DeviceModel(engine, mdl)
        res = {}
        for name, flags in (("default", 0), ("f32", B.RUN_F32_RNN)):
            b = B.Batch(dm, 1, sig.size)
            b.set_signals(sig[None, :])
            b.run(1.0, flags); b.finish()
            res[name] = (b.basecall(0), b.path(0)[0], b.transitions(0))
            b.close()
        dm.close()
        d = float(np.abs(res["default"][2] - ref["trans"]).max())
        apart = int((res["default"][1] != ref["path"]).sum())
        print("near tie %d (%s): %d samples; default path %d bases, f32 path %d, oracle %d; |dtrans| %.2e; Viterbi paths apart in %d blocks"
              % (i, str(g["source%d" % i]), sig.size, len(res["default"][0]), len(res["f32"][0]), len(ref["basecall"]), d, apart))
        assert d <= float(g["dtrans_bound%d" % i])
        assert apart <= int(g["blocks_apart%d" % i])
        if int(g["f32_equals_oracle%d" % i]) == 1:
            assert res["f32"][0] == ref["basecall"]
