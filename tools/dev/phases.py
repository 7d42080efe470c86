#!/usr/bin/env python3
"""Where a step of the layer kernel goes, per wave role, in the kernel AS IT RUNS IN PRODUCTION (round 5): the -DFFHIP_PHASES variant of libffhip.so
(tools/dev/build_variants.sh phases="-DFFHIP_PHASES", copied over the tree's library by the caller) adds the time between its stamps to per-wave words
in LDS and hands the sums out at the end of each launch.  usage: tools/dev/phases.py [config=c2|h256|c4|h512] [pairs=6] [serial]   (serial: one batch at a time, run + finish -- a launch alone on the chip)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B  # noqa: E402
from flappie_amd import model as M  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
serial = len(sys.argv) > 3 and sys.argv[3] == "serial"
KIND, H, NREAD, T, PAIR = {"c2": (M.NET_LSTM5, 384, 256, 4000, True), "h256": (M.NET_LSTM5, 256, 1024, 4000, False),
                           "c4": (M.NET_GRUMOD5, 256, 1024, 4000, False), "h512": (M.NET_LSTM5, 512, 256, 20000, False)}[cfg]
L = B.lib()
L.ffhip_debug_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
L.ffhip_debug_phases.restype = C.c_int
eng = B.Engine(0)
dm = B.DeviceModel(eng, M.synthetic_model(KIND, H, seed=1))
rng = np.random.default_rng(1)
bs = [B.Batch(dm, NREAD, T) for _ in range(2)]
for b in bs:
    b.set_signals(rng.standard_normal((NREAD, T)).astype(np.float32))
buf = (C.c_ulonglong * 72)()
assert L.ffhip_debug_phases(buf, 1) == 0          # arms the buffer
for rnd in range(rounds + 1):
    if serial:
        for b in bs:
            b.run(); b.finish()
    elif PAIR:
        bs[0].run_pair(bs[1])
    else:
        bs[0].run(); bs[1].run()
    for b in bs:
        b.finish()
    if rnd == 0:
        L.ffhip_debug_phases(buf, 1)              # warm-up round dropped
L.ffhip_debug_phases(buf, 0)
v = np.array(list(buf), dtype=np.float64)
cyc, steps = v[:64].reshape(8, 8), v[64:72]
names = ["turn", "poll", "matrix", "bar1", "gates", "bar2"]
print("config %s: cycles per step by wave (mean over workgroups and steps; counter ticks)  -- waves 0-3 x, 4-7 h" % cfg)
print("wave  " + "".join("%9s" % n for n in names) + "    total")
for w in range(8):
    if steps[w] > 0:
        per = cyc[w, :6] / steps[w]
        print("%4d  " % w + "".join("%9.0f" % x for x in per) + "%9.0f" % per.sum())
for b in bs:
    b.close()
dm.close(); eng.close()
