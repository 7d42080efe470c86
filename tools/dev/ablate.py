#!/usr/bin/env python3
"""Layer-kernel ablation timings in the REAL engine.  Build one library per variant with
    (apply tools/dev/experiments/lstm_split_ablation_switches.patch first: the switches are not in the product source)
    make -C flappie_amd/csrc CXXFLAGS="... -DFFHIP_SPLIT_ABLATE=<bits>" ffhip_rnn_split.o -B && make -C flappie_amd/csrc
(bits: 1 no projection MFMAs, 2 no hand-off wait, 4 no gate math, 8 no sweep of h, 16 no prefetch of x), copy it over
flappie_amd/libffhip.so on the GPU box and run this script: it prints the time of the five recurrent layers of the
headline batch.  The variants compute wrong results by construction; only the timing is of interest.  (Run-time switches or
in-kernel timestamps change the compiler's wait-count placement and slow the kernel by ~35 %: compile-time variants do not.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import binding as B, model as M
label = sys.argv[1] if len(sys.argv) > 1 else "0"
eng = B.Engine(0)
mdl = M.synthetic_model(M.NET_LSTM5, 384, seed=1)
dm = B.DeviceModel(eng, mdl)
sig = np.random.default_rng(1).standard_normal((256, 4000)).astype(np.float32)
b = B.Batch(dm, 256, 4000)
b.set_signals(sig)
try:
    for _ in range(3):
        b.run(1.0, B.RUN_NO_DECODE); b.finish()
    eng.set_profiling(True)
    b.run(1.0, B.RUN_NO_DECODE); b.finish()
    p = b.profile()
    print("variant %s: recurrent %.3f ms per 5 layers = %.0f cycles per step" % (label, p["recurrent"]["ms"], p["recurrent"]["ms"] / 4000 * 2.4e6))
except Exception as e:
    print("variant", label, "failed:", e)
