#!/usr/bin/env python3
"""Where does a packed launch whose members re-sweep on purpose (-DFFHIP_FORCE_RETRY=2 variant) first differ from the plain library?  One 1024-read GRUmod batch (H = 256), every
layer's output kept (FFHIP_RUN_KEEP_ACTS), a few reads of the first two tile pairs.  usage: tools/dev/retry_diff.py dump OUT.npz [kind=gru|lstm]   |   tools/dev/retry_diff.py cmp A.npz B.npz"""
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
READS = [0, 5, 15, 16, 21, 31, 32, 47, 48, 63]

if sys.argv[1] == "dump":
    from flappie_amd import binding as B
    from flappie_amd import model as M
    kind = M.NET_LSTM5 if (len(sys.argv) > 3 and sys.argv[3] == "lstm") else M.NET_GRUMOD5
    eng = B.Engine(0)
    dm = B.DeviceModel(eng, M.synthetic_model(kind, 256, seed=5 + kind))
    rng = np.random.default_rng(7)
    nread, T = 1024, 800
    b = B.Batch(dm, nread, T)
    b.set_signals(rng.standard_normal((nread, T)).astype(np.float32))
    b.run(1.0, B.RUN_KEEP_ACTS); b.finish()
    out = {"path": np.int64(b.rnn_path())}
    for l in range(-1, 5):
        for r in READS:
            out["act_%d_%d" % (l, r)] = b.activation(l, r)
    np.savez(sys.argv[2], **out)
    print("wrote", sys.argv[2], "rnn path", b.rnn_path(), "blocks", b.nblock)
else:
    a, c = np.load(sys.argv[2]), np.load(sys.argv[3])
    for l in range(-1, 5):
        for r in READS:
            x, y = a["act_%d_%d" % (l, r)], c["act_%d_%d" % (l, r)]
            d = np.abs(x - y)
            if d.max() > 0:
                blocks = np.nonzero(d.max(axis=1))[0]
                first = blocks[0] if l % 2 == 0 else blocks[0]
                units = np.nonzero(d[blocks[0]])[0]
                unitsl = np.nonzero(d[blocks[-1]])[0]
                print("layer output %d read %2d (tile %d): %d of %d blocks differ, first %d (units %s%s), last %d (units %s%s), max %.2e"
                      % (l, r, r // 16, len(blocks), x.shape[0], blocks[0], units[:12].tolist(), "..." if len(units) > 12 else "", blocks[-1], unitsl[:12].tolist(),
                         "..." if len(unitsl) > 12 else "", d.max()))
            else:
                print("layer output %d read %2d (tile %d): equal" % (l, r, r // 16))
