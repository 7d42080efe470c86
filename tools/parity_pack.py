#!/usr/bin/env python3
"""Parity of the packed layer kernels (H = 256, 1024 reads a launch: k_lstm_pack, k_grumod_pack) against the oracle: a ragged 1024-read
batch of random reads per model through the engine, every read through the oracle (a process pool: the oracle is one thread a read),
counting base-string / quality / path mismatches and the largest transition-score difference.  Run on the GPU box.
usage: tools/parity_pack.py [nread=1024] [max samples=1500]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

_om = None


def _init(kind, H, seed):
    global _om
    from oracle import ffo
    _om = ffo.OracleModel(M.synthetic_model(kind, H, seed=seed))


def _call(x):
    r = _om.basecall(x)
    return dict(basecall=r["basecall"], quality=r["quality"], path=np.asarray(r["path"]), trans=np.asarray(r["trans"]))


def main():
    nread = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    tmax = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    t0 = time.time()
    work = []
    for kind, H, seed in ((M.NET_LSTM5, 256, 6), (M.NET_GRUMOD5, 256, 7)):
        rng = np.random.default_rng(200 + seed)
        lens = np.sort(rng.integers(300, tmax + 1, nread))[::-1]
        sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
        with mp.Pool(min(48, os.cpu_count() or 1), initializer=_init, initargs=(kind, H, seed)) as pool:      # (before the engine exists: no HIP state is forked)
            refs = pool.map(_call, sigs, chunksize=4)
        work.append((kind, H, seed, sigs, refs))
        print("oracle: kind %d H %d, %d reads, %.0f s" % (kind, H, nread, time.time() - t0), flush=True)
    from flappie_amd import binding as B
    eng = B.Engine(0)
    tot = dict(reads=0, bases=0, base_mismatch=0, qual_mismatch=0, path_mismatch=0, qual_chars_diff=0, worst=0.0)
    for kind, H, seed, sigs, refs in work:
        dm = B.DeviceModel(eng, M.synthetic_model(kind, H, seed=seed))
        assert dm.launch_reads == nread or nread != 1024, dm.launch_reads
        b = B.Batch(dm, nread, max(x.size for x in sigs))
        b.set_signals_ragged(sigs)
        b.run(); b.finish()
        for r, ref in enumerate(refs):
            tot["reads"] += 1
            tot["bases"] += len(ref["basecall"])
            tot["worst"] = max(tot["worst"], float(np.abs(b.transitions(r) - ref["trans"]).max()))
            if b.basecall(r) != ref["basecall"]:
                tot["base_mismatch"] += 1
            elif b.quality(r) != ref["quality"]:
                tot["qual_mismatch"] += 1
                tot["qual_chars_diff"] += sum(1 for a, c in zip(b.quality(r), ref["quality"]) if a != c)
            if not np.array_equal(b.path(r)[0], ref["path"]):
                tot["path_mismatch"] += 1
        b.close(); dm.close()
        print("kind %d H %d done (%.0f s): %s" % (kind, H, time.time() - t0, tot), flush=True)
    print("campaign:", tot)


if __name__ == "__main__":
    main()
