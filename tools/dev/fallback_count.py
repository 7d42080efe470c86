#!/usr/bin/env python3
"""How often a layer launch takes its re-sweep path (a sweep of h(t-1) that met a sentinel is done again): the -DFFHIP_COUNT_FALLBACK variant of libffhip.so counts them in the
word next to the batch's abort word.  Per workload: pairs / batches as bench.py runs them.  usage: tools/dev/fallback_count.py [rounds=6]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import binding as B, model as M  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eng = B.Engine(0)
L = B.lib()
L.ffhip_debug_batch_counter.argtypes = [C.c_void_p]
L.ffhip_debug_batch_counter.restype = C.c_uint
for name, kind, H, nread, T, pair, wg_waves in (("c2 (pairs)", M.NET_LSTM5, 384, 256, 4000, True, 256 * 4), ("h256", M.NET_LSTM5, 256, 1024, 4000, False, 512 * 4),
                                                ("c4", M.NET_GRUMOD5, 256, 1024, 4000, False, 512 * 4)):
    dm = B.DeviceModel(eng, M.synthetic_model(kind, H, seed=1))
    rng = np.random.default_rng(1)
    bs = [B.Batch(dm, nread, T) for _ in range(2)]
    for b in bs:
        b.set_signals(rng.standard_normal((nread, T)).astype(np.float32))
    n0 = sum(L.ffhip_debug_batch_counter(b.h) for b in bs)
    for _ in range(rounds):
        if pair:
            bs[0].run_pair(bs[1])
        else:
            bs[0].run(); bs[1].run()
        for b in bs:
            b.finish()
    n1 = sum(L.ffhip_debug_batch_counter(b.h) for b in bs)
    nblock = bs[0].nblock
    wave_steps = rounds * 2 * 5 * (nblock - 1) * wg_waves          # rounds x batches x layers x steps x (workgroups x h waves) of a batch
    print("%-12s re-sweeps: %d of %d h-wave steps = %.4f %%  (one per %.0f layer launches)" % (name, n1 - n0, wave_steps, 100.0 * (n1 - n0) / wave_steps,
                                                                                                 (rounds * 2 * 5) / max(1, n1 - n0)))
    for b in bs:
        b.close()
    dm.close()
eng.close()
