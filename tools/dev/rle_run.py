import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from flappie_amd import binding as B, model as M
eng = B.Engine(0)
dm = B.DeviceModel(eng, M.synthetic_model(M.NET_LSTM5_RLE, 384, seed=3))
sig = np.random.default_rng(5).standard_normal((256, 4000)).astype(np.float32)
b = B.Batch(dm, 256, 4000); b.set_signals(sig)
for _ in range(4):
    b.run(); b.finish()
import time
t0 = time.time()
for _ in range(5):
    b.run(); b.finish()
dt = (time.time() - t0) / 5
print("run-length model H 384, 256 x 4000: %.2f ms per batch = %.1f Msamples/s" % (dt * 1e3, 256 * 4000 / dt / 1e6))
