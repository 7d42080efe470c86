#!/usr/bin/env python3
"""Quick GPU-side parity + timing probe (development tool; the real tests live in tests/)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B  # noqa: E402
from flappie_amd import model as M  # noqa: E402
from oracle import ffo  # noqa: E402


def parity(kind, hidden, T, nread, eng, flags=0, seed=7):
    mdl = M.synthetic_model(kind, hidden, seed=seed)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(eng, mdl)
    rng = np.random.default_rng(123 + T)
    sig = rng.standard_normal((nread, T)).astype(np.float32)
    b = B.Batch(dm, nread, T)
    b.set_signals(sig)
    b.run(1.0, flags | B.RUN_KEEP_ACTS)
    b.finish()
    worst = 0.0
    nbad = 0
    for r in range(nread):
        ref = om.basecall(sig[r], viterbi_only=bool(flags & B.RUN_VITERBI_ONLY))
        tr = b.transitions(r)
        d_tr = np.abs(tr - ref["trans"]).max()
        worst = max(worst, d_tr)
        path, qpath = b.path(r)
        same_path = np.array_equal(path, ref["path"])
        bc, q = b.basecall(r), b.quality(r)
        msg = "read %d: |dtrans| %.2e path %s basecall %s (len %d/%d) qual %s score %.4f/%.4f" % (
            r, d_tr, same_path, bc == ref["basecall"], len(bc), len(ref["basecall"]), q == ref["quality"],
            b.score(r), ref["score"])
        if not (flags & B.RUN_VITERBI_ONLY):
            d_po = np.abs(b.posterior(r) - ref["post"]).max()
            trc = b.trace(r)
            msg += " |dpost| %.2e trace maxdiff %d" % (d_po, np.abs(trc - ref["trace"]).max())
        print(msg)
        if bc != ref["basecall"]:
            nbad += 1
    print("kind %d H %d T %d nread %d: worst |dtrans| %.3e, basecall mismatches %d" % (kind, hidden, T, nread, worst, nbad))
    b.close()
    dm.close()
    return worst, nbad


def timing(eng, hidden=384, nread=256, T=4000, steps=3):
    mdl = M.synthetic_model(M.NET_LSTM5, hidden, seed=1)
    dm = B.DeviceModel(eng, mdl)
    rng = np.random.default_rng(1)
    sig = rng.standard_normal((nread, T)).astype(np.float32)
    b = B.Batch(dm, nread, T)
    b.set_signals(sig)
    b.run(); b.finish()
    t0 = time.time()
    for _ in range(steps):
        b.run(); b.finish()
    dt = (time.time() - t0) / steps
    print("H %d B %d T %d: %.2f ms/batch = %.2f Msamples/s" % (hidden, nread, T, dt * 1e3, nread * T / dt / 1e6))
    eng.set_profiling(True)
    b.run(); b.finish()
    prof = b.profile()
    for k, v in prof.items():
        print("   %-18s %8.3f ms  %5d launches" % (k, v["ms"], v["launches"]))
    eng.set_profiling(False)
    print("   basecall[0][:60] =", b.basecall(0)[:60])
    b.close(); dm.close()


if __name__ == "__main__":
    eng = B.Engine(0)
    print(eng.info())
    what = sys.argv[1:] or ["parity", "timing"]
    if "parity" in what:
        parity(M.NET_LSTM5, 64, 4000, 5, eng)
        parity(M.NET_LSTM5, 96, 1237, 3, eng)
        parity(M.NET_GRUMOD5, 64, 2000, 4, eng)
        parity(M.NET_LSTM5, 64, 1003, 2, eng, flags=B.RUN_VITERBI_ONLY)
        parity(M.NET_LSTM5, 36, 601, 17, eng)
    if "nread" in what:
        for nr in (64, 128, 256):
            timing(eng, 384, nr, 4000, steps=2)
    if "timing" in what:
        timing(eng, 384, 256, 4000)


def timing_nread():
    eng = B.Engine(0)
    for nread in (64, 128, 256):
        timing(eng, 384, nread, 4000, steps=2)
