#!/usr/bin/env python3
"""Parity campaign at bench.py's headline shape THROUGH THE PAIRED LAUNCH: H = 384 LSTM (bench.py's model,
synthetic_model(seed=1)), batches of 256 reads run two at a time through ffhip_batch_run_pair -- k_lstm_split_pair<0,3,2,true>, the kernel
the driver times.  Every read also goes through the oracle TWICE (process pools, started before the engine exists):

  oracle       dot products float, term by term in index order (ff_oracle.c dot mode 0: what the parity tests use)
  oracle+blas  the same algorithm with its GEMV / GEMM calls -- the reference's cblas_sgemv / cblas_sgemm shapes, layers.c:1009,
               flappie_matrix.c:384 -- in a real OpenBLAS dlopen()ed from this host (dot mode 3): the arithmetic the reference
               itself would do on this machine

so that the engine's distance from the oracle stands beside the distance BETWEEN TWO SUMMATION ORDERS OF THE REFERENCE ALGORITHM on
the same reads: a called base that flips between those two is a near-tie of the posterior decode, not a property of the engine.
Half the pairs are uniform (every read `tmax` samples), half ragged (1500 .. tmax, sorted as the flappie binary sorts).
Run on the GPU box.   usage: tools/parity_h384.py [nread=2048] [tmax=2500] [shape=c2]
shape: c2 = LSTM H 384 in pairs of 256-read batches (the default, the headline); h256 / c4 = LSTM / GRUmod H 256 in full 1024-read launches of the
packed kernels (k_lstm_pack / k_grumod_pack); c5 = LSTM H 512 in 256-read batches (k_lstm_split<0,4,2>).  bench.py's models (seed 1)."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

SHAPES = {"c2": (M.NET_LSTM5, 384, 256, True), "h256": (M.NET_LSTM5, 256, 1024, False), "c4": (M.NET_GRUMOD5, 256, 1024, False), "c5": (M.NET_LSTM5, 512, 256, False)}
SHAPE = sys.argv[3] if len(sys.argv) > 3 else "c2"
KIND, H, PER_BATCH, PAIRED = SHAPES[SHAPE]
SEED = 1
_om = None


def _init(mode):
    global _om
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle import ffo
    _om = ffo.OracleModel(M.synthetic_model(KIND, H, seed=SEED))
    got, _ = ffo.use_dot_mode(mode)
    assert got == mode, "no LP64 OpenBLAS on this host"


def _call(x):
    r = _om.basecall(x)
    return dict(basecall=r["basecall"], quality=r["quality"], path=np.asarray(r["path"]), trans=np.asarray(r["trans"]))


class Tally:
    def __init__(self, what):
        self.what = what
        self.d = dict(reads=0, bases=0, base_mismatch=0, bases_apart=0, qual_mismatch=0, qual_chars_diff=0, path_mismatch=0, worst=0.0)
        self.named = []

    def add(self, tag, a, ref):
        d = self.d
        d["reads"] += 1
        d["bases"] += len(ref["basecall"])
        dt = float(np.abs(a["trans"] - ref["trans"]).max())
        d["worst"] = max(d["worst"], dt)
        if a["basecall"] != ref["basecall"]:
            d["base_mismatch"] += 1
            import difflib
            sm = difflib.SequenceMatcher(None, a["basecall"], ref["basecall"], autojunk=False)
            apart = max(len(a["basecall"]), len(ref["basecall"])) - sum(m.size for m in sm.get_matching_blocks())
            d["bases_apart"] += apart
            self.named.append("%s: %d bases against %d (%d apart), |dtrans| %.2e" % (tag, len(a["basecall"]), len(ref["basecall"]), apart, dt))
        elif a["quality"] != ref["quality"]:
            d["qual_mismatch"] += 1
            d["qual_chars_diff"] += sum(1 for x, y in zip(a["quality"], ref["quality"]) if x != y)
        if not np.array_equal(a["path"], ref["path"]):
            d["path_mismatch"] += 1

    def line(self):
        d = self.d
        return ("%-24s %d reads, %d bases: %d reads with another base string (%d bases apart in all = %.1f per million), %d more with another quality string "
                "(%d characters), %d with another Viterbi path; worst |dtrans| %.2e"
                % (self.what + ":", d["reads"], d["bases"], d["base_mismatch"], d["bases_apart"], 1e6 * d["bases_apart"] / max(1, d["bases"]), d["qual_mismatch"],
                   d["qual_chars_diff"], d["path_mismatch"], d["worst"]))


def main():
    nread = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    tmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
    assert nread % (2 * PER_BATCH) == 0
    t0 = time.time()
    rng = np.random.default_rng(3840)
    batches = []
    for k in range(nread // PER_BATCH):
        if (k // 2) % 2 == 0 and SHAPE == "c2" or SHAPE != "c2" and k % 2 == 0:
            sigs = [rng.standard_normal(tmax).astype(np.float32) for _ in range(PER_BATCH)]
        else:
            lens = np.sort(rng.integers(tmax * 3 // 5, tmax + 1, PER_BATCH))[::-1]
            sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
        batches.append(sigs)
    flat = [x for sigs in batches for x in sigs]
    ncpu = min(128, os.cpu_count() or 1)
    with mp.Pool(ncpu, initializer=_init, initargs=(0,)) as pool:
        ref0 = pool.map(_call, flat, chunksize=2)
    print("oracle: %d reads, %d samples, %.0f s" % (nread, sum(x.size for x in flat), time.time() - t0), flush=True)
    with mp.Pool(ncpu, initializer=_init, initargs=(3,)) as pool:
        ref3 = pool.map(_call, flat, chunksize=2)
    from oracle import ffo
    print("oracle+blas (%s): %.0f s" % (ffo.use_dot_mode(3)[1], time.time() - t0), flush=True)
    ffo.use_dot_mode(0)
    from flappie_amd import binding as B
    eng = B.Engine(0)
    dm = B.DeviceModel(eng, M.synthetic_model(KIND, H, seed=SEED))
    t_eng0, t_eng3, t_00 = Tally("engine <-> oracle"), Tally("engine <-> oracle+blas"), Tally("oracle <-> oracle+blas")
    min_kmers, paired = 10 ** 9, 0
    bs = [B.Batch(dm, PER_BATCH, tmax) for _ in range(2)]
    for k in range(0, len(batches), 2):
        for j in (0, 1):
            bs[j].set_signals_ragged(batches[k + j])
        if PAIRED:
            bs[0].run_pair(bs[1])
        else:
            bs[0].run(); bs[1].run()
        for j in (0, 1):
            b = bs[j]
            b.finish()
            paired += int(b.paired())
            assert b.rnn_path() == 3
            for r in range(PER_BATCH):
                i = (k + j) * PER_BATCH + r
                a = dict(basecall=b.basecall(r), quality=b.quality(r), path=b.path(r)[0], trans=b.transitions(r))
                tag = "batch %d read %d (%d samples)" % (k + j, r, flat[i].size)
                t_eng0.add(tag, a, ref0[i])
                t_eng3.add(tag, a, ref3[i])
                t_00.add(tag, ref0[i], ref3[i])
                s = ref0[i]["basecall"]
                min_kmers = min(min_kmers, len({s[q:q + 5] for q in range(len(s) - 4)}))
        print("pair %d done (%.0f s)" % (k // 2, time.time() - t0), flush=True)
    for b in bs:
        b.close()
    dm.close()
    print("campaign: shape %s (kind %d, H = %d, %d reads a batch%s), %d of %d batches in a paired launch; %d reads, %d samples; fewest distinct 5-mers in a read %d"
          % (SHAPE, KIND, H, PER_BATCH, ", through ffhip_batch_run_pair" if PAIRED else "", paired, len(batches), nread, sum(x.size for x in flat), min_kmers))
    for t in (t_eng0, t_eng3, t_00):
        print(t.line())
        for n in t.named:
            print("     " + n)
    both = set(n.split(":")[0] for n in t_eng0.named) & set(n.split(":")[0] for n in t_eng3.named)
    print("reads the engine calls differently from BOTH evaluations of the reference algorithm: %d%s" % (len(both), (" -- " + "; ".join(sorted(both))) if both else ""))


if __name__ == "__main__":
    main()
