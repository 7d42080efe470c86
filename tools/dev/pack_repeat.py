#!/usr/bin/env python3
"""Does a packed layer launch give the same bits every time?  (round 5: one run of tests/test_split_gpu.py::test_dense_launch_matches_the_one_tile_launches[1-1040]
failed with a development variant of the packed kernels and passed the next 18.)  The test's ragged 1040-read batch, run REPS times per cell kind; every read's
scores, calls and qualities against the first run's and against the one-tile launches (FFHIP_DEBUG=no_dense).  usage: tools/dev/pack_repeat.py [reps=200] [kinds=GRUmod,LSTM]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B  # noqa: E402
from flappie_amd import model as M  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["GRUmod", "LSTM"]
reffile = sys.argv[3] if len(sys.argv) > 3 else None      # the reference (one-tile launches) of ANOTHER library: written if absent, read if present
eng = B.Engine(0)
for kind, name in ((M.NET_GRUMOD5, "GRUmod"), (M.NET_LSTM5, "LSTM")):
    if name not in kinds:
        continue
    nread, T = 1040, 1000
    dm = B.DeviceModel(eng, M.synthetic_model(kind, 256, seed=5 + kind))
    rng = np.random.default_rng(nread)
    lens = rng.integers(200, T + 1, size=nread)
    lens[:3] = (T, 200, 237)
    sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]

    def once():
        b = B.Batch(dm, nread, T)
        b.set_signals_ragged(sigs)
        b.run(); b.finish()
        out = [(b.transitions(r), b.basecall(r), b.quality(r)) for r in range(nread)]
        b.close()
        return out
    import pickle
    rf = (reffile + "." + name) if reffile else None
    if rf and os.path.exists(rf):
        ref = pickle.load(open(rf, "rb"))
    else:
        os.environ["FFHIP_DEBUG"] = "no_dense"
        ref = once()
        os.environ.pop("FFHIP_DEBUG")
        if rf:
            pickle.dump(ref, open(rf, "wb"))
    nbad = 0
    for rep in range(reps):
        got = once()
        bad = [r for r in range(nread) if not (np.array_equal(got[r][0], ref[r][0]) and got[r][1:] == ref[r][1:])]
        if bad:
            nbad += 1
            first = [int(np.nonzero(np.abs(got[r][0] - ref[r][0]).max(axis=1))[0][0]) if np.abs(got[r][0] - ref[r][0]).any() else -1 for r in bad]
            print("%s rep %d: %d reads differ from the one-tile launches: reads %s (tiles %s), max |dtrans| %.2e, first differing blocks %s, lengths %s"
                  % (name, rep, len(bad), bad[:20], sorted(set(r // 16 for r in bad))[:8], max(float(np.abs(got[r][0] - ref[r][0]).max()) for r in bad), first[:8],
                     [int(lens[r]) for r in bad[:8]]), flush=True)
    print("%s, H = 256, %d ragged reads: %d of %d runs deviate" % (name, nread, nbad, reps), flush=True)
    dm.close()
eng.close()
