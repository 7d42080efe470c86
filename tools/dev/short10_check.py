import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from flappie_amd import binding as B, model as M
from oracle import ffo
eng = B.Engine(0)
for kind, H in ((M.NET_GRUMOD5, 64), (M.NET_GRUMOD5, 128)):
    mdl = M.synthetic_model(kind, H, seed=9)
    om = ffo.OracleModel(mdl); dm = B.DeviceModel(eng, mdl)
    rng = np.random.default_rng(4)
    lens = [19, 20, 21, 22, 25, 33, 35, 37, 64, 100, 0, 127, 128, 129, 130, 131, 255, 257, 4096 + 3, 4096 * 2 + 1, 5000]
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    b = B.Batch(dm, len(sigs), max(lens)); b.set_signals_ragged(sigs); b.run(); b.finish()
    bad = 0; worst = 0.0
    for r, n in enumerate(lens):
        if not n: continue
        ref = om.basecall(sigs[r])
        ok = b.basecall(r) == ref["basecall"] and b.quality(r) == ref["quality"] and np.array_equal(b.path(r)[0], ref["path"][:len(b.path(r)[0])])
        worst = max(worst, float(np.abs(b.posterior(r) - ref["post"]).max()))
        bad += (not ok)
    print("kind", kind, "H", H, "reads", len(lens), "bad", bad, "worst |dpost|", worst)
