#!/usr/bin/env python3
"""Which read of tools/parity_pack.py's GRUmod batch (1024 reads of up to 3000 samples) is called differently from the oracle, by how much
its scores differ, and what the other layer-kernel paths of the engine say about it (one-tile / dense forms: FFHIP_DEBUG=no_pack in a second
process; the f32-input MFMA kernel: RUN_F32_RNN).  Development tool; run on the GPU box."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import parity_pack as P  # noqa: E402
from flappie_amd import model as M  # noqa: E402


def main():
    kind, H, seed, nread, tmax = M.NET_GRUMOD5, 256, 7, 1024, 3000
    rng = np.random.default_rng(200 + seed)
    lens = np.sort(rng.integers(300, tmax + 1, nread))[::-1]
    sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
    with mp.Pool(48, initializer=P._init, initargs=(kind, H, seed)) as pool:
        refs = pool.map(P._call, sigs, chunksize=4)
    from flappie_amd import binding as B
    eng = B.Engine(0)
    dm = B.DeviceModel(eng, M.synthetic_model(kind, H, seed=seed))
    out = {}
    for tag, flags in (("default", 0), ("f32", B.RUN_F32_RNN)):
        b = B.Batch(dm, nread, max(x.size for x in sigs))
        b.set_signals_ragged(sigs)
        b.run(1.0, flags); b.finish()
        out[tag] = [(b.basecall(r), b.transitions(r).copy(), b.path(r)[0].copy()) for r in range(nread)]
        b.close()
    print("layer kernel path:", "packed" if not "no_pack" in os.environ.get("FFHIP_DEBUG", "") else "FFHIP_DEBUG=no_pack (dense / one-tile forms)")
    for r, ref in enumerate(refs):
        d = out["default"][r]
        if d[0] != ref["basecall"]:
            f = out["f32"][r]
            dt = np.abs(d[1] - ref["trans"]).max(axis=1)
            blk = np.flatnonzero(d[2] != ref["path"])
            print("read %d (%d samples, %d blocks): default path calls %d bases, oracle %d; max |dtrans| %.3g (block %d); f32 path equals oracle: %s, equals default: %s"
                  % (r, sigs[r].size, ref["trans"].shape[0], len(d[0]), len(ref["basecall"]), dt.max(), int(dt.argmax()), f[0] == ref["basecall"], f[0] == d[0]))
            print("   Viterbi paths differ in blocks %s" % (blk[:12],))
            k = int(blk[0]) if blk.size else 0
            # the posterior margin at the first differing block: best and runner-up of the oracle's transition posterior are not available here; show the scores' margin instead
            row_o, row_d = np.sort(ref["trans"][max(0, k - 1)])[-3:], np.sort(d[1][max(0, k - 1)])[-3:]
            print("   three largest transition scores at block %d: oracle %s, engine %s" % (max(0, k - 1), row_o, row_d))
    import hashlib
    print("digest of all default-path transition scores:", hashlib.sha256(b"".join(x[1].tobytes() for x in out["default"])).hexdigest()[:16])


if __name__ == "__main__":
    main()
