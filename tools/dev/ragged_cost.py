"""host cost of setting up a ragged batch (apply_lengths + uploads) by reads per batch and hidden size"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import model as M, binding as B
eng = B.Engine(0)
rng = np.random.default_rng(1)
for hidden, nread in ((384, 256), (256, 256), (256, 512), (384, 512)):
    dm = B.DeviceModel(eng, M.synthetic_model(M.NET_LSTM5, hidden, seed=1))
    b = B.Batch(dm, nread, 5600)
    sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in sorted(rng.integers(3200, 5200, nread), reverse=True)]
    ts = []
    for it in range(5):
        t0 = time.perf_counter(); b.set_signals_ragged(sigs); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); b.run(); t1 = time.perf_counter(); b.finish(); t2 = time.perf_counter()
    print("H %d, %d reads: set_signals_ragged %s ms; run (enqueue) %.2f ms, finish %.2f ms" % (hidden, nread, " ".join("%.2f" % (t * 1e3) for t in ts), (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    b.close(); dm.close()
