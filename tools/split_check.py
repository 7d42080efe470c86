#!/usr/bin/env python3
"""Development probe for the split-bf16 recurrent kernel (ffhip_rnn_split.hip): parity against the oracle at
H = 128, agreement with the f32-MFMA kernel at the headline shape, and timing of both."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B  # noqa: E402
from flappie_amd import model as M  # noqa: E402
from oracle import ffo  # noqa: E402


def parity(eng, hidden, T, nread, seed=7):
    mdl = M.synthetic_model(M.NET_LSTM5, hidden, seed=seed)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(eng, mdl)
    rng = np.random.default_rng(123 + T)
    sig = rng.standard_normal((nread, T)).astype(np.float32)
    out = {}
    for name, flags in (("split", 0), ("f32", B.RUN_F32_RNN)):
        b = B.Batch(dm, nread, T)
        b.set_signals(sig)
        b.run(1.0, flags)
        b.finish()
        worst, nbad = 0.0, 0
        for r in range(nread):
            ref = om.basecall(sig[r])
            worst = max(worst, float(np.abs(b.transitions(r) - ref["trans"]).max()))
            nbad += (b.basecall(r) != ref["basecall"]) + (b.quality(r) != ref["quality"])
        out[name] = (worst, nbad)
        print("H %d T %d nread %d %-5s: worst |dtrans| vs oracle %.3e, basecall/quality mismatches %d" % (hidden, T, nread, name, worst, nbad), flush=True)
        b.close()
    dm.close()
    return out


def compare(eng, hidden=384, nread=256, T=4000, steps=3):
    mdl = M.synthetic_model(M.NET_LSTM5, hidden, seed=1)
    dm = B.DeviceModel(eng, mdl)
    rng = np.random.default_rng(1)
    sig = rng.standard_normal((nread, T)).astype(np.float32)
    res = {}
    for name, flags in (("split", 0), ("f32", B.RUN_F32_RNN)):
        b = B.Batch(dm, nread, T)
        b.set_signals(sig)
        b.run(1.0, flags); b.finish()
        t0 = time.time()
        for _ in range(steps):
            b.run(1.0, flags); b.finish()
        dt = (time.time() - t0) / steps
        print("%-5s H %d B %d T %d: %.2f ms/batch = %.2f Msamples/s" % (name, hidden, nread, T, dt * 1e3, nread * T / dt / 1e6), flush=True)
        eng.set_profiling(True)
        b.run(1.0, flags); b.finish()
        for k, v in b.profile().items():
            print("   %-18s %8.3f ms  %5d launches" % (k, v["ms"], v["launches"]))
        eng.set_profiling(False)
        res[name] = ([b.transitions(r) for r in range(nread)], [b.basecall(r) for r in range(nread)], [b.quality(r) for r in range(nread)])
        b.close()
    d = max(float(np.abs(x - y).max()) for x, y in zip(res["split"][0], res["f32"][0]))
    nb = sum(x != y for x, y in zip(res["split"][1], res["f32"][1]))
    nq = sum(x != y for x, y in zip(res["split"][2], res["f32"][2]))
    print("split vs f32 kernel: max |dtrans| %.3e, basecall differences %d / %d reads, quality differences %d" % (d, nb, nread, nq))
    dm.close()


if __name__ == "__main__":
    eng = B.Engine(0)
    print(eng.info())
    what = sys.argv[1:] or ["parity", "compare"]
    if "parity" in what:
        parity(eng, 128, 1500, 40)
        parity(eng, 128, 603, 7)
    if "compare" in what:
        compare(eng)
    if "h256" in what:
        compare(eng, hidden=256)
