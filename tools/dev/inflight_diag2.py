"""which co-running kernel disturbs batch 0's posterior?  batch 1 is run with different recurrent paths / models."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import model as M, binding as B

nread, T = int(os.environ.get("NREAD", "512")), 1500
eng = B.Engine(0)
mdlA = M.synthetic_model(M.NET_LSTM5, int(os.environ.get("HA", "256")), seed=3)
dmA = B.DeviceModel(eng, mdlA)
rng = np.random.default_rng(11)
sigA = rng.standard_normal((nread, T)).astype(np.float32)
sigB = rng.standard_normal((nread, T)).astype(np.float32)

def grab(b):
    return [b.posterior(r) for r in range(nread)]

b = B.Batch(dmA, nread, T); b.set_signals(sigA); b.run(); b.finish(); alone = grab(b); b.close()
cases = [("same model, default path", mdlA, 0), ("same model, f32 persistent kernels", mdlA, B.RUN_F32_RNN),
         ("same model, launch-per-step kernels", mdlA, B.RUN_STEPWISE_RNN),
         ("H = 384 model, default path", M.synthetic_model(M.NET_LSTM5, 384, seed=4), 0),
         ("GRUmod H = 256, default path", M.synthetic_model(M.NET_GRUMOD5, 256, seed=5), 0),
         ("H = 96 model (f32 path)", M.synthetic_model(M.NET_LSTM5, 96, seed=6), 0)]
if os.environ.get("QUICK"): cases = cases[:1]
for name, mdlB, flagsB in cases:
    dmB = dmA if mdlB is mdlA else B.DeviceModel(eng, mdlB)
    b0, b1 = B.Batch(dmA, nread, T), B.Batch(dmB, nread, T)
    nbad, cols = 0, {}
    for rnd in range(6):
        b0.set_signals(sigA); b0.run()
        b1.set_signals(sigB); b1.run(1.0, flagsB)
        b0.finish(); b1.finish()
        g = grab(b0)
        for r in range(nread):
            if not np.array_equal(g[r], alone[r]):
                nbad += 1
                d = np.abs(g[r].astype(np.float64) - alone[r])
                blk = int(np.nonzero(d.max(axis=1) > 0)[0][0])
                e = int(d[blk].argmax())
                cols[(blk % 64, e)] = cols.get((blk % 64, e), 0) + 1
    print("%-40s corrupted reads of batch 0 in 6 rounds: %d   (first block %% 64, entry) -> count: %s" % (name, nbad, dict(sorted(cols.items())[:8])))
    b0.close(); b1.close()
    if dmB is not dmA: dmB.close()
