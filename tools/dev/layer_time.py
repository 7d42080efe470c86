#!/usr/bin/env python3
"""Time of the five recurrent layers of the headline batch with whatever libffhip.so is in place (for interleaved A/B
comparisons of library variants on ONE device: devices of the pool differ by up to 4 % on the split layer kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import binding as B, model as M
eng = B.Engine(0)
mdl = M.synthetic_model(M.NET_LSTM5, 384, seed=1)
dm = B.DeviceModel(eng, mdl)
sig = np.random.default_rng(1).standard_normal((256, 4000)).astype(np.float32)
b = B.Batch(dm, 256, 4000)
b.set_signals(sig)
for _ in range(3):
    b.run(); b.finish()
eng.set_profiling(True)
ms = []
for _ in range(3):
    b.run(); b.finish()
    ms.append(b.profile()["recurrent"]["ms"])
print("%s recurrent %.3f ms (min of 3: %s)" % (sys.argv[1] if len(sys.argv) > 1 else "", min(ms), " ".join("%.3f" % x for x in ms)))
