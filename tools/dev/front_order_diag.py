#!/usr/bin/env python3
"""Which order / round / batch of tests/test_bench_shapes_gpu.py::test_front_order_changes_the_schedule_not_the_results deviates, and how (round 5: the test
failed about once in ten runs).  usage: tools/dev/front_order_diag.py [repeats=30] [var=FFHIP_DEBUG|FFHIP_FRONT_ORDER]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B  # noqa: E402
from flappie_amd import model as M  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
var = sys.argv[2] if len(sys.argv) > 2 else "FFHIP_DEBUG"
orders = sys.argv[3].split(",") if len(sys.argv) > 3 else ["layers", "batch", "none"]
eng = B.Engine(0)
dm = B.DeviceModel(eng, M.synthetic_model(M.NET_LSTM5, 384, seed=1))
rng = np.random.default_rng(4004)
T, nread = 1500, 256
sigs = [rng.standard_normal((nread, T)).astype(np.float32) for _ in range(4)]
batches = [B.Batch(dm, nread, T) for _ in range(4)]
ref = {}
nbad = 0
for rep in range(reps):
    for order in orders:
        os.environ[var] = ("front_order=" + order) if var == "FFHIP_DEBUG" else order
        for rnd in range(3):
            for k in range(4):
                batches[k].set_signals(sigs[(k + rnd) % 4])
            batches[0].run_pair(batches[1])
            batches[2].run_pair(batches[3])
            for k, b in enumerate(batches):
                b.finish()
                sig = (k + rnd) % 4
                got = [(b.transitions(r), b.basecall(r), b.quality(r)) for r in range(nread)]
                if sig not in ref:
                    ref[sig] = got
                    continue
                bad = [r for r in range(nread) if not (np.array_equal(got[r][0], ref[sig][r][0]) and got[r][1] == ref[sig][r][1] and got[r][2] == ref[sig][r][2])]
                if bad:
                    nbad += 1
                    d = [float(np.abs(got[r][0] - ref[sig][r][0]).max()) for r in bad]
                    first = [int(np.nonzero(np.abs(got[r][0] - ref[sig][r][0]).max(axis=1))[0][0]) if np.abs(got[r][0] - ref[sig][r][0]).any() else -1 for r in bad]
                    print("rep %d order %s round %d batch object %d (signal set %d, paired %s): %d reads differ: reads %s, max |dtrans| %s, first differing block %s"
                          % (rep, order, rnd, k, sig, b.paired(), len(bad), bad[:12], ["%.2e" % x for x in d[:12]], first[:12]), flush=True)
print("%d repeats x orders %s x 3 rounds x 4 batches: %d deviating batches" % (reps, orders, nbad))
