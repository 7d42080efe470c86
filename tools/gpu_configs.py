#!/usr/bin/env python3
"""Run the other BASELINE.json configurations once: shape support, sanity properties, timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B, model as M
from oracle import ffo


def run(eng, kind, H, nread, T, check_reads=(0,), label=""):
    mdl = M.synthetic_model(kind, H, seed=3)
    dm = B.DeviceModel(eng, mdl)
    sig = np.random.default_rng(5).standard_normal((nread, T)).astype(np.float32)
    b = B.Batch(dm, nread, T)
    b.set_signals(sig)
    b.run(); b.finish()
    t0 = time.time(); b.run(); b.finish(); dt = time.time() - t0
    eng.set_profiling(True); b.run(); b.finish(); prof = b.profile(); eng.set_profiling(False)
    msg = "%s kind %d H %d nread %d T %d nblock %d: %.1f ms = %.2f Msamples/s | " % (label, kind, H, nread, T, b.nblock, dt * 1e3, nread * T / dt / 1e6)
    msg += " ".join("%s %.2f" % (k, v["ms"]) for k, v in prof.items())
    print(msg, flush=True)
    om = ffo.OracleModel(mdl)
    for r in check_reads:
        t1 = time.time()
        ref = om.basecall(sig[r])
        d = np.abs(b.transitions(r) - ref["trans"]).max()
        print("    read %d vs oracle (%.0f s): |dtrans| %.2e basecall %s quality %s len %d" % (
            r, time.time() - t1, d, b.basecall(r) == ref["basecall"], b.quality(r) == ref["quality"], len(ref["basecall"])), flush=True)
    b.close(); dm.close()


def run_rle(eng):
    mdl = M.synthetic_model(M.NET_LSTM5_RLE, 384, seed=3)
    dm = B.DeviceModel(eng, mdl)
    sig = np.random.default_rng(5).standard_normal((256, 4000)).astype(np.float32)
    b = B.Batch(dm, 256, 4000)
    b.set_signals(sig)
    b.run(); b.finish()
    t0 = time.time(); b.run(); b.finish(); dt = time.time() - t0
    eng.set_profiling(True); b.run(); b.finish(); prof = b.profile(); eng.set_profiling(False)
    print("runnie-shape H 384 256 x 4000: %.1f ms = %.2f Msamples/s | " % (dt * 1e3, 256 * 4000 / dt / 1e6) + " ".join("%s %.2f" % (k, v["ms"]) for k, v in prof.items()), flush=True)
    b.close(); dm.close()


if __name__ == "__main__":
    eng = B.Engine(0)
    what = sys.argv[1:] or ["c4", "h256", "h512", "c5"]
    if "h256" in what: run(eng, M.NET_LSTM5, 256, 256, 4000, label="r941_native(20200220)-shape")
    if "c4" in what: run(eng, M.NET_GRUMOD5, 256, 256, 4000, label="C4 r941_5mC-shape")
    if "h512" in what: run(eng, M.NET_LSTM5, 512, 256, 4000, label="r103_native-shape")
    if "c5" in what: run(eng, M.NET_LSTM5, 512, 16, 100000, check_reads=(), label="C5 long reads (16)")
    if "c5big" in what: run(eng, M.NET_LSTM5, 512, 256, 100000, check_reads=(), label="C5 long reads (256)")
    if "rle" in what: run_rle(eng)
    if "nread" in what:
        for n in (256, 384, 512):
            run(eng, M.NET_GRUMOD5, 256, n, 4000, check_reads=(), label="C4 nread %d" % n)
            run(eng, M.NET_LSTM5, 256, n, 4000, check_reads=(), label="H256 nread %d" % n)
        for n in (256, 512):
            run(eng, M.NET_LSTM5, 384, n, 4000, check_reads=(), label="H384 nread %d" % n)
