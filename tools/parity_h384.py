#!/usr/bin/env python3
"""Parity campaign at bench.py's headline shape THROUGH THE PAIRED LAUNCH: H = 384 LSTM (bench.py's model,
synthetic_model(seed=1)), batches of 256 reads run two at a time through ffhip_batch_run_pair -- k_lstm_split_pair<0,3,2,true>, the kernel
the driver times -- every read through the oracle (a process pool, started before the engine exists).  Half the pairs are uniform
(every read `tmax` samples), half ragged (1500 .. tmax, sorted as the flappie binary sorts).  Counts base-string / quality-string /
Viterbi-path mismatches and the largest transition-score difference; reads recorded in tests/golden/near_ties.npz would be named.
Run on the GPU box.   usage: tools/parity_h384.py [nread=2048] [tmax=2500]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

KIND, H, SEED = M.NET_LSTM5, 384, 1
_om = None


def _init():
    global _om
    from oracle import ffo
    _om = ffo.OracleModel(M.synthetic_model(KIND, H, seed=SEED))


def _call(x):
    r = _om.basecall(x)
    return dict(basecall=r["basecall"], quality=r["quality"], path=np.asarray(r["path"]), trans=np.asarray(r["trans"]))


def main():
    nread = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    tmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
    assert nread % 512 == 0
    t0 = time.time()
    rng = np.random.default_rng(3840)
    batches = []
    for k in range(nread // 256):
        if (k // 2) % 2 == 0:
            sigs = [rng.standard_normal(tmax).astype(np.float32) for _ in range(256)]
        else:
            lens = np.sort(rng.integers(1500, tmax + 1, 256))[::-1]
            sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
        batches.append(sigs)
    with mp.Pool(min(128, os.cpu_count() or 1), initializer=_init) as pool:
        refs = [pool.map(_call, sigs, chunksize=2) for sigs in batches]
    print("oracle: %d reads, %d samples, %.0f s" % (nread, sum(x.size for s in batches for x in s), time.time() - t0), flush=True)
    from flappie_amd import binding as B
    eng = B.Engine(0)
    dm = B.DeviceModel(eng, M.synthetic_model(KIND, H, seed=SEED))
    tot = dict(reads=0, samples=0, bases=0, base_mismatch=0, qual_mismatch=0, path_mismatch=0, qual_chars_diff=0, worst=0.0, min_kmers=10 ** 9, paired_batches=0)
    bs = [B.Batch(dm, 256, tmax) for _ in range(2)]
    for k in range(0, len(batches), 2):
        for j in (0, 1):
            bs[j].set_signals_ragged(batches[k + j])
        bs[0].run_pair(bs[1])
        for j in (0, 1):
            b = bs[j]
            b.finish()
            tot["paired_batches"] += int(b.paired())
            for r, ref in enumerate(refs[k + j]):
                tot["reads"] += 1
                tot["samples"] += batches[k + j][r].size
                tot["bases"] += len(ref["basecall"])
                s = ref["basecall"]
                tot["min_kmers"] = min(tot["min_kmers"], len({s[i:i + 5] for i in range(len(s) - 4)}))
                d = float(np.abs(b.transitions(r) - ref["trans"]).max())
                tot["worst"] = max(tot["worst"], d)
                if b.basecall(r) != ref["basecall"]:
                    tot["base_mismatch"] += 1
                    print("  batch %d read %d (%d samples): %d bases against the oracle's %d, |dtrans| %.2e" % (k + j, r, batches[k + j][r].size, len(b.basecall(r)), len(s), d))
                elif b.quality(r) != ref["quality"]:
                    tot["qual_mismatch"] += 1
                    tot["qual_chars_diff"] += sum(1 for a, c in zip(b.quality(r), ref["quality"]) if a != c)
                if not np.array_equal(b.path(r)[0], ref["path"]):
                    tot["path_mismatch"] += 1
        print("pair %d done (%.0f s): %s" % (k // 2, time.time() - t0, tot), flush=True)
    for b in bs:
        b.close()
    dm.close()
    print("campaign (H = 384, run_pair, %d of %d batches in a paired launch):" % (tot["paired_batches"], len(batches)), tot)


if __name__ == "__main__":
    main()
