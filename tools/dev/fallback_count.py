import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import binding as B, model as M
eng = B.Engine(0)
mdl = M.synthetic_model(M.NET_LSTM5, 384, seed=1)
dm = B.DeviceModel(eng, mdl)
sig = np.random.default_rng(1).standard_normal((256, 4000)).astype(np.float32)
b = B.Batch(dm, 256, 4000)
b.set_signals(sig)
L = B.lib(); L.ffhip_debug_batch_counter.argtypes = [C.c_void_p]; L.ffhip_debug_batch_counter.restype = C.c_uint
n0 = L.ffhip_debug_batch_counter(b.h)
for _ in range(4):
    b.run(); b.finish()
n1 = L.ffhip_debug_batch_counter(b.h)
steps = 4 * 5 * 799 * 256 * 4          # runs x layers x steps x workgroups x h waves
print("re-sweeps: %d of %d wave-steps = %.4f %%" % (n1 - n0, steps, 100.0 * (n1 - n0) / steps))
