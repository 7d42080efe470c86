"""which of {alone, two in flight} disagrees with the oracle, and where (posterior / qpath / quality)?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import model as M, binding as B
from oracle import ffo

kind, hidden, nread, T = M.NET_LSTM5, 256, 512, 1500
mdl = M.synthetic_model(kind, hidden, seed=3)
eng = B.Engine(0)
dm = B.DeviceModel(eng, mdl)
rng = np.random.default_rng(hidden + nread)
sig = [rng.standard_normal((nread, T)).astype(np.float32) for _ in range(2)]

def grab(b):
    return [dict(trans=b.transitions(r), post=b.posterior(r), path=b.path(r)[0], qpath=b.path(r)[1], base=b.basecall(r), qual=b.quality(r), trace=b.trace(r)) for r in range(nread)]

def diff(a, c, tag, other=None):
    bad = [r for r in range(nread) if a[r]["qual"] != c[r]["qual"] or a[r]["base"] != c[r]["base"]]
    print(tag, "reads with different strings:", bad[:20], len(bad))
    for r in bad[:3]:
        for k in ("trans", "post", "qpath", "trace"):
            x, y = np.asarray(a[r][k], dtype=np.float64), np.asarray(c[r][k], dtype=np.float64)
            d = np.abs(x - y); d[np.isnan(d)] = 0
            print("   read", r, k, "max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "shape", d.shape)
        print("   path equal", np.array_equal(a[r]["path"], c[r]["path"]))
        d = np.abs(np.asarray(a[r]["post"], dtype=np.float64) - c[r]["post"]).max(axis=1)
        blks = np.nonzero(d > 0)[0]
        print("   blocks whose posterior differs:", blks.tolist()[:80], "n =", int((d > 0).sum()))
        if other is not None and blks.size:
            print("   ... and there the in-flight posterior equals the OTHER signal's posterior of that slot:", np.array_equal(c[r]["post"][blks], other[r]["post"][blks]),
                  "| all zero:", not np.any(c[r]["post"][blks]), "| equals trans:", np.array_equal(c[r]["post"][blks], c[r]["trans"][blks]))
            dd = (np.asarray(c[r]["post"], dtype=np.float64) - a[r]["post"])
            for bk in (blks[0], blks[-1]):
                print("   block", bk, "diff by entry (rows = to 0..3 | flop row):")
                print(np.array2string(dd[bk][:32].reshape(4, 8), precision=4, suppress_small=True), np.array2string(dd[bk][32:40], precision=4, suppress_small=True))
            print("   in flight", c[r]["post"][blks[0]][:6], "alone", a[r]["post"][blks[0]][:6], "other", other[r]["post"][blks[0]][:6])
    return bad

alone = []
for k in range(2):
    b = B.Batch(dm, nread, T); b.set_signals(sig[k]); b.run(); b.finish(); alone.append(grab(b)); b.close()
again = []
for k in range(2):
    b = B.Batch(dm, nread, T); b.set_signals(sig[k]); b.run(); b.finish(); again.append(grab(b)); b.close()
diff(alone[0], again[0], "alone vs alone again, signals 0:")
diff(alone[1], again[1], "alone vs alone again, signals 1:")
bs = [B.Batch(dm, nread, T) for _ in range(2)]
allbad = set()
for rnd in range(3):
    for k in range(2):
        bs[k].set_signals(sig[(k + rnd) % 2]); bs[k].run()
    for k in range(2):
        bs[k].finish()
        allbad |= set((( k + rnd) % 2, r) for r in diff(alone[(k + rnd) % 2], grab(bs[k]), "round %d batch %d in flight vs alone:" % (rnd, k), alone[1 - (k + rnd) % 2]))
# the same two batch objects, one at a time
for k in range(2):
    bs[k].set_signals(sig[k]); bs[k].run(); bs[k].finish()
    diff(alone[k], grab(bs[k]), "reused batch %d, one at a time, vs alone:" % k)
om = ffo.OracleModel(mdl)
for (s, r) in sorted(allbad)[:4]:
    ref = om.basecall(sig[s][r])
    print("oracle read", s, r, "alone matches oracle:", alone[s][r]["qual"] == ref["quality"], alone[s][r]["base"] == ref["basecall"],
          "max |dpost|", np.abs(alone[s][r]["post"] - ref["post"]).max())
