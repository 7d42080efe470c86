#!/usr/bin/env python3
"""Time of the five recurrent layers of a 1024-read GRUmod batch (H = 256, 4000 samples) with whatever libffhip.so is in place
(interleaved A/B runs of library variants on one device: tools/dev/build_variants.sh, tools/dev/ab_run.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import binding as B, model as M
nread = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = B.Engine(0)
dm = B.DeviceModel(eng, M.synthetic_model(M.NET_GRUMOD5, 256, seed=1))
sig = np.random.default_rng(1).standard_normal((nread, 4000)).astype(np.float32)
b = B.Batch(dm, nread, 4000)
b.set_signals(sig)
for _ in range(3):
    b.run(); b.finish()
eng.set_profiling(True)
ms = []
for _ in range(4):
    b.run(); b.finish()
    ms.append(b.profile()["recurrent"]["ms"])
print("%s recurrent %.3f ms (min of 4: %s)" % (sys.argv[1] if len(sys.argv) > 1 else "", min(ms), " ".join("%.3f" % x for x in ms)))
