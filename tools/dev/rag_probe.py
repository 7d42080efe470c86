import sys, numpy as np
sys.path.insert(0, '/root/repo')
from flappie_amd import binding as B, model as M
eng = B.Engine(0)
variant = sys.argv[1]
mdl = M.synthetic_model(M.NET_LSTM5, 64 if 'h64' in variant else 128, seed=11)
dm = B.DeviceModel(eng, mdl)
rng = np.random.default_rng(5)
lens = [1200, 1199, 600, 601, 37, 19, 1000, 800, 801, 802, 803, 804, 805, 806, 807, 808, 300, 1200, 45]
if 'long' in variant:
    lens = [1200] * 19
sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
E = np.zeros(0, dtype=np.float32)
lead = 16 if 'lead16' in variant else 17
flags = B.RUN_F32_RNN
if 'nodecode' in variant: flags |= B.RUN_NO_DECODE
if 'stepwise' in variant: flags |= B.RUN_STEPWISE_RNN
b = B.Batch(dm, 40, 1200)
b.set_signals_ragged([E] * lead + sigs + [E] * (40 - lead - len(sigs)))
b.run(1.0, flags); b.finish()
print(variant, "ok", flush=True)
