import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import binding as B, model as M
eng = B.Engine(0)
for H in (256,):
    mdl = M.synthetic_model(M.NET_LSTM5, H, seed=1)
    dm = B.DeviceModel(eng, mdl)
    sig = np.random.default_rng(1).standard_normal((256, 4000)).astype(np.float32)
    b = B.Batch(dm, 256, 4000)
    b.set_signals(sig)
    for name, fl in (("fused split", 0), ("GEMM + recurrence-only", B.RUN_UNFUSED_RNN)):
        for _ in range(3):
            b.run(1.0, fl); b.finish()
        eng.set_profiling(True)
        b.run(1.0, fl); b.finish()
        p = b.profile()
        eng.set_profiling(False)
        print("H %d %-24s path %d: inproj %.2f recurrent %.2f ms" % (H, name, b.rnn_path(), p["inproj"]["ms"], p["recurrent"]["ms"]))
