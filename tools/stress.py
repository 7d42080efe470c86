#!/usr/bin/env python3
"""Soak test of the engine on the GPU box: many uniform and ragged batches of the benchmark's shape through the persistent
recurrent kernels -- TWO BATCHES IN FLIGHT, as bench.py and the flappie binary run them (argv[4] = 1 for one) -- checking for
errors / timeouts and that the results of a fixed probe read, placed in a random slot of every batch, never change.
usage: tools/stress.py [iterations] [kind] [hidden] [batches in flight] [reads per batch] [pair]
`pair` = 1: the batches go out two at a time through ffhip_batch_run_pair (batches in flight: 4 = two pairs, as bench.py's default)."""
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
nfl = int(sys.argv[4]) if len(sys.argv) > 4 else 2
nread = int(sys.argv[5]) if len(sys.argv) > 5 else 256
pair = len(sys.argv) > 6 and sys.argv[6] == "1"
if pair and nfl % 2:
    sys.exit("pairs need an even number of batches in flight (a batch waits for its partner before it is run: ADVICE r4)")
import ctypes as C  # noqa: E402
eng = B.Engine(0)
B.lib().ffhip_debug_fallback_count.argtypes = [C.c_void_p]
mdl = M.synthetic_model(kind, hidden, seed=1)
dm = B.DeviceModel(eng, mdl)
rng = np.random.default_rng(0)
probe = rng.standard_normal(3777).astype(np.float32)
batches = [B.Batch(dm, nread, 6000) for _ in range(nfl)]
slots = [0] * nfl
ref = None
t0 = time.time()
nsamp = 0
nbad = 0


def check(k, it):
    global ref
    b, slot = batches[k], slots[k]
    b.finish()
    got = (b.basecall(slot), b.quality(slot), b.transitions(slot).tobytes(), b.path(slot)[0].tobytes())
    if ref is None:
        ref = got
    if got != ref:
        what = [n for n, x, y in zip(("bases", "quality", "transitions", "path"), got, ref) if x != y]
        d = np.abs(np.frombuffer(got[2], dtype=np.float32) - np.frombuffer(ref[2], dtype=np.float32)) if len(got[2]) == len(ref[2]) else None
        global nbad
        nbad += 1
        if nbad <= 8:
            print("MISMATCH: probe read changed in the batch checked at iteration %d (slot %d, batch object %d): %s differ; max |dtrans| %s, differing blocks %s" % (
                it, slot, k, what, None if d is None else float(d.max()), None if d is None or not d.any() else sorted(set((np.nonzero(d)[0] // 40).tolist()))[:12]), flush=True)


pending = []
for it in range(niter):
    k = it % nfl
    if len(pending) == nfl:
        check(pending.pop(0), it)
    mode = it % 3
    if mode == 0:
        lens = np.full(nread, 6000)
    elif mode == 1:
        lens = np.sort(rng.integers(2000, 6000, nread))[::-1]
    else:
        lens = rng.integers(19, 6000, nread)                  # unsorted, extreme spread
    lens = lens.copy()
    slots[k] = int(rng.integers(0, nread))
    lens[slots[k]] = probe.size
    sigs = [probe if i == slots[k] else rng.standard_normal(int(n)).astype(np.float32) for i, n in enumerate(lens)]
    nsamp += int(lens.sum())
    batches[k].set_signals_ragged(sigs)
    if not pair:
        batches[k].run()
    elif k % 2 == 1:
        batches[k - 1].run_pair(batches[k], 1.0, 0)       # (uniform lengths of equal capacity pair up; the engine decides)
    elif it == niter - 1:
        batches[k].run()
    pending.append(k)
    if it % 250 == 0:
        print("iteration %d ok (%.1f s)" % (it, time.time() - t0), flush=True)
for k in pending:
    check(k, niter)
print(("stress ok" if nbad == 0 else "STRESS FAILED (%d mismatching probes)" % nbad) + ": %d batches of %d reads (%s H=%d, %d in flight), %.1f Gsamples, %.1f s, fallbacks to the launch-per-step kernels: %d" % (
    niter, nread, "LSTM" if kind == M.NET_LSTM5 else "GRUmod", hidden, nfl, nsamp / 1e9, time.time() - t0, B.lib().ffhip_debug_fallback_count(eng.h)))
