#!/usr/bin/env python3
"""Soak test of the engine on the GPU box: many uniform and ragged batches of the benchmark's shape through the
persistent recurrent kernels, checking for errors/timeouts and that results of a fixed probe read never change."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B  # noqa: E402
from flappie_amd import model as M  # noqa: E402

niter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kind = int(sys.argv[2]) if len(sys.argv) > 2 else M.NET_LSTM5
hidden = int(sys.argv[3]) if len(sys.argv) > 3 else 384
eng = B.Engine(0)
mdl = M.synthetic_model(kind, hidden, seed=1)
dm = B.DeviceModel(eng, mdl)
rng = np.random.default_rng(0)
probe = rng.standard_normal(3777).astype(np.float32)
b = B.Batch(dm, 256, 6000)
ref = None
t0 = time.time()
for it in range(niter):
    mode = it % 3
    if mode == 0:
        lens = np.full(256, 6000)
    elif mode == 1:
        lens = np.sort(rng.integers(2000, 6000, 256))[::-1]
    else:
        lens = rng.integers(19, 6000, 256)                    # unsorted, extreme spread
    lens = lens.copy()
    slot = int(rng.integers(0, 256))
    lens[slot] = probe.size
    sigs = [probe if i == slot else rng.standard_normal(int(n)).astype(np.float32) for i, n in enumerate(lens)]
    b.set_signals_ragged(sigs)
    b.run(); b.finish()
    got = (b.basecall(slot), b.quality(slot), b.transitions(slot).tobytes(), b.path(slot)[0].tobytes())
    if ref is None:
        ref = got
    assert got == ref, "probe read changed in iteration %d (slot %d, mode %d)" % (it, slot, mode)
    if it % 10 == 0:
        print("iteration %d ok (%.1f s)" % (it, time.time() - t0), flush=True)
print("stress ok: %d batches, %.1f s" % (niter, time.time() - t0))
