#!/usr/bin/env python3
"""Development check of packed batches: every read of a packed batch against the same read in a one-read-a-row (ragged) batch, bit for bit.
usage: tools/dev/packed_check.py [kind=lstm|gru] [hidden=128] [nrow=32] [cap=4000] [nread=80]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B, model as M

kind = sys.argv[1] if len(sys.argv) > 1 else "lstm"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nrow = int(sys.argv[3]) if len(sys.argv) > 3 else 32
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
nread = int(sys.argv[5]) if len(sys.argv) > 5 else 80
rng = np.random.default_rng(5)
eng = B.Engine(0)
dm = B.DeviceModel(eng, M.synthetic_model(M.NET_GRUMOD5 if kind == "gru" else M.NET_LSTM5, H, seed=1))
lens = np.clip(np.exp(np.log(cap / 5) + 0.8 * rng.standard_normal(nread)), 200, cap - 50).astype(int)
sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
pb = B.Batch(dm, nrow, cap, max_reads=nread)
slot, off = pb.pack_plan([x.size for x in sigs])
keep = [i for i in range(nread) if slot[i] >= 0]
print("%d of %d reads placed in %d rows of %d samples; gap %d blocks; fill %.2f" % (len(keep), nread, nrow, cap, B.lib().ffhip_model_pack_gap(dm.h),
      sum(sigs[i].size for i in keep) / (nrow * cap)))
pb.set_signals_packed([sigs[i] for i in keep], [slot[i] for i in keep], [off[i] for i in keep])
t0 = time.time(); pb.run(); pb.finish(); print("packed run %.3f s, rnn path %d, reads %d" % (time.time() - t0, pb.rnn_path(), pb.nreads()))
bad = 0
for k0 in range(0, len(keep), nrow):
    grp = keep[k0:k0 + nrow]
    ub = B.Batch(dm, len(grp), cap)
    ub.set_signals_ragged([sigs[i] for i in grp])
    ub.run(); ub.finish()
    for j, i in enumerate(grp):
        v = k0 + j
        same = (pb.basecall(v) == ub.basecall(j) and pb.quality(v) == ub.quality(j) and np.array_equal(pb.transitions(v), ub.transitions(j))
                and np.array_equal(pb.posterior(v), ub.posterior(j)) and np.array_equal(pb.path(v)[0], ub.path(j)[0])
                and np.array_equal(pb.path(v)[1][1:], ub.path(j)[1][1:]) and np.array_equal(pb.trace(v), ub.trace(j)) and pb.score(v) == ub.score(j))
        if not same:
            bad += 1
            dt = np.abs(pb.transitions(v) - ub.transitions(j))
            print("read %d (row %d, block %d, %d samples): differs; max |dtrans| %.3g first bad block %s, strings equal %s" % (i, slot[i], off[i], sigs[i].size, dt.max(),
                  np.argwhere(dt.max(axis=1) > 0)[:3].ravel(), pb.basecall(v) == ub.basecall(j)))
    ub.close()
print("%d of %d reads differ from their one-read-a-row evaluation" % (bad, len(keep)))
sys.exit(1 if bad else 0)
