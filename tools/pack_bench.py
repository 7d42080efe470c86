#!/usr/bin/env python3
"""Packed batches at the engine's boundary, signals resident in HBM (no files, no host pipeline): what a batch of MIXED read lengths delivers once its reads stand
several to a row -- beside the same rows filled with equal reads one to a row (include/ffhip.h "packed batches"; the binary's side of it: tools/length_mix.py).

A window of log-normal read lengths (median 8000, sigma 1, clipped to 1000 .. `longest`) is planned into `rows` rows of max(longest, total / rows) samples
(ffhip_pack_plan: longest first, each into the emptiest row; FFHIP_DEBUG=pack_first_fit: the rule before round 6's third session) -- the flappie binary's policy --, two batch objects in flight, K steps each; reported: Msamples/s of REAL samples, the
fill of the rows (samples / (rows x the longest row)), the HIP-event time of the layer launches per step.  The uniform leg is `rows` reads of the row length.
Run on the GPU box.   usage: tools/pack_bench.py [hidden=384] [rows=512] [longest=60000] [steps=6]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B, model as M  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 384
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 512
longest = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
eng = B.Engine(0)
dm = B.DeviceModel(eng, M.synthetic_model(M.NET_LSTM5, H, seed=1))
rng = np.random.default_rng(20260930)
# a window whose samples fill `rows` rows of the longest read about once
lens = []
while sum(lens) < rows * longest:
    lens.append(int(np.clip(np.exp(np.log(8000) + rng.standard_normal()), 1000, longest)))
lens.sort(reverse=True)
cap = ((max(lens[0] + 64, int(sum(lens) / rows * 1.05) + 64 * int(B.lib().ffhip_model_pack_gap(dm.h))) + 1023) // 1024) * 1024
noise = rng.standard_normal(cap + 8).astype(np.float32)
sigs = [noise[:n] for n in lens]                      # (values do not matter for the time; one buffer keeps the host side small)


def timed(batches, label, real_samples, row_samples):
    eng.set_profiling(True)
    for b in batches:                                  # warm-up: the lazily allocated buffers, the first launches
        b.run()
    for b in batches:
        b.finish()
    eng.synchronize()
    t0 = time.perf_counter()
    pending = []
    for i in range(steps):
        b = batches[i % len(batches)]
        if len(pending) == len(batches):
            pending.pop(0).finish()
        b.run()
        pending.append(b)
    for b in pending:
        b.finish()
    eng.synchronize()
    dt = time.perf_counter() - t0
    layer_ms = float(np.mean([b.profile()["recurrent"]["ms"] for b in batches]))
    print("%-8s %d steps in %.3f s: %.1f Msamples/s of real samples; rows %d x %d samples, fill %.3f; layer launches %.1f ms a step (%.2f us a block-step of the longest row); rnn path %d"
          % (label, steps, dt, steps * real_samples / dt / 1e6, rows, cap, real_samples / row_samples, layer_ms, 1e3 * layer_ms / 5 / (row_samples / rows / (cap / batches[0].nblock)), batches[0].rnn_path()), flush=True)      # (a layer launch runs the longest ROW's blocks, not the capacity's: the packed leg's figure divided by the capacity until round 6's third session and read ~4 % low)
    eng.set_profiling(False)
    return steps * real_samples / dt / 1e6


print("H = %d, %d rows; window of %d reads, %.1f Msamples (log-normal: median 8000, sigma 1, 1000 .. %d); row capacity %d samples" % (H, rows, len(lens), sum(lens) / 1e6, longest, cap), flush=True)
pbs = [B.Batch(dm, rows, cap, max_reads=len(lens)) for _ in range(2)]
slot, off = pbs[0].pack_plan(lens)
keep = [i for i in range(len(lens)) if slot[i] >= 0]
for pb in pbs:
    pb.set_signals_packed([sigs[i] for i in keep], [slot[i] for i in keep], [off[i] for i in keep])
real = sum(lens[i] for i in keep)
row_end = {}
for i in keep:
    row_end[slot[i]] = max(row_end.get(slot[i], 0), off[i] * round(cap / pbs[0].nblock) + lens[i])
longest_row = max(row_end.values())
print("planned: %d of %d reads placed, longest row %d samples" % (len(keep), len(lens), longest_row), flush=True)
v_mixed = timed(pbs, "packed", real, rows * longest_row)
for pb in pbs:
    pb.close()
ubs = [B.Batch(dm, rows, longest_row) for _ in range(2)]
u = np.tile(noise[:longest_row], (rows, 1))
for ub in ubs:
    ub.set_signals(u)
v_uni = timed(ubs, "uniform", rows * longest_row, rows * longest_row)
print("packed / uniform = %.3f" % (v_mixed / v_uni))
