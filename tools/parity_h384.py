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
Run on the GPU box.   usage: tools/parity_h384.py [nread=2048] [tmax=2500] [shape=c2] [--make-refs FILE | --refs FILE] [--gates default,fast] [--trans-reads N]
  --make-refs FILE   no GPU: both oracle evaluations of every read, base / quality strings and a checksum of the Viterbi path only (a few MB), written to FILE --
                     run where CPU time is free (the build container: same libff_oracle.so, same OpenBLAS binary and kernel family; the GPU box re-derives
                     --trans-reads of them and insists on identical strings before it uses the file)
  --refs FILE        take the oracle's strings from FILE; the transition scores are compared on the first --trans-reads reads (default 512), which
                     the box evaluates itself
  --gates a,b        engine passes: default = the reference's exp_ps / division replayed bit for bit, fast = FFHIP_RUN_FAST_GATES (hardware v_exp / v_rcp), fast2 = FFHIP_RUN_FAST_GATES2 (two-word exponent, Newton step; the library's default since round 6), exact = default here
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
_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
_opt = {}
_argv = sys.argv[1:]
_pos = []
while _argv:
    a = _argv.pop(0)
    if a.startswith("--"):
        _opt[a[2:]] = _argv.pop(0)
    else:
        _pos.append(a)
SHAPE = _pos[2] if len(_pos) > 2 else "c2"
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


def _path_sum(p):
    import zlib
    return zlib.crc32(np.ascontiguousarray(p, dtype=np.int32).tobytes())


def _call_small(x):
    r = _om.basecall(x)
    return (r["basecall"], r["quality"], _path_sum(r["path"]))


class Tally:
    def __init__(self, what):
        self.what = what
        self.d = dict(reads=0, bases=0, base_mismatch=0, bases_apart=0, qual_mismatch=0, qual_chars_diff=0, path_mismatch=0, worst=0.0)
        self.named = []

    def add(self, tag, a, ref):
        d = self.d
        d["reads"] += 1
        d["bases"] += len(ref["basecall"])
        dt = float(np.abs(a["trans"] - ref["trans"]).max()) if (a.get("trans") is not None and ref.get("trans") is not None) else float("nan")
        if dt == dt:
            d["worst"] = max(d["worst"], dt)
            d["trans_reads"] = d.get("trans_reads", 0) + 1
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
        if a["psum"] != ref["psum"]:
            d["path_mismatch"] += 1

    def line(self):
        d = self.d
        return ("%-24s %d reads, %d bases: %d reads with another base string (%d bases apart in all = %.1f per million), %d more with another quality string "
                "(%d characters), %d with another Viterbi path; worst |dtrans| %.2e over %d reads"
                % (self.what + ":", d["reads"], d["bases"], d["base_mismatch"], d["bases_apart"], 1e6 * d["bases_apart"] / max(1, d["bases"]), d["qual_mismatch"],
                   d["qual_chars_diff"], d["path_mismatch"], d["worst"], d.get("trans_reads", 0)))


def _small(r):
    return dict(basecall=r["basecall"], quality=r["quality"], psum=_path_sum(r["path"]), trans=r["trans"])


def main():
    nread = int(_pos[0]) if len(_pos) > 0 else 2048
    tmax = int(_pos[1]) if len(_pos) > 1 else 2500
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
    ncpu = int(_opt.get("procs", min(128, os.cpu_count() or 1)))
    from oracle import ffo
    if "make-refs" in _opt:                      # ---- no GPU: strings and path checksums of every read, both evaluations
        out = {}
        for mode in (0, 3):
            with mp.Pool(ncpu, initializer=_init, initargs=(mode,)) as pool:
                res = pool.map(_call_small, flat, chunksize=2)
            out["base%d" % mode] = np.array([r[0] for r in res])
            out["qual%d" % mode] = np.array([r[1] for r in res])
            out["psum%d" % mode] = np.array([r[2] for r in res], dtype=np.uint32)
            print("mode %d: %d reads, %.0f s" % (mode, nread, time.time() - t0), flush=True)
        np.savez_compressed(_opt["make-refs"], shape=SHAPE, nread=nread, tmax=tmax, blas=str(ffo.use_dot_mode(3)[1]), **out)
        return
    ntrans = nread
    stored = None
    if "refs" in _opt:
        stored = np.load(_opt["refs"])
        assert str(stored["shape"]) == SHAPE and int(stored["nread"]) == nread and int(stored["tmax"]) == tmax, "the reference file was made for another campaign"
        ntrans = min(nread, int(_opt.get("trans-reads", 512)))
    with mp.Pool(ncpu, initializer=_init, initargs=(0,)) as pool:
        ref0 = [_small(r) for r in pool.map(_call, flat[:ntrans], chunksize=2)]
    print("oracle: %d reads, %d samples, %.0f s" % (ntrans, sum(x.size for x in flat[:ntrans]), time.time() - t0), flush=True)
    with mp.Pool(ncpu, initializer=_init, initargs=(3,)) as pool:
        ref3 = [_small(r) for r in pool.map(_call, flat[:ntrans], chunksize=2)]
    print("oracle+blas (%s): %.0f s" % (ffo.use_dot_mode(3)[1], time.time() - t0), flush=True)
    ffo.use_dot_mode(0)
    if stored is not None:
        # the file's strings must be what this host computes where it computes them at all; then the rest of the reads come from the file
        for mode, ref in ((0, ref0), (3, ref3)):
            bad = [i for i in range(ntrans) if ref[i]["basecall"] != str(stored["base%d" % mode][i]) or ref[i]["quality"] != str(stored["qual%d" % mode][i])
                   or ref[i]["psum"] != int(stored["psum%d" % mode][i])]
            print("reference file against this host's oracle, mode %d: %d of %d reads differ%s" % (mode, len(bad), ntrans, (" -- " + str(bad[:8])) if bad else ""), flush=True)
            if mode == 0:
                assert not bad, "the stored oracle strings are not this tree's oracle"
            for i in range(ntrans, nread):
                ref.append(dict(basecall=str(stored["base%d" % mode][i]), quality=str(stored["qual%d" % mode][i]), psum=int(stored["psum%d" % mode][i]), trans=None))
    from flappie_amd import binding as B
    eng = B.Engine(0)
    dm = B.DeviceModel(eng, M.synthetic_model(KIND, H, seed=SEED))
    t_00 = Tally("oracle <-> oracle+blas")
    for i in range(nread):
        t_00.add("read %d" % i, ref0[i], ref3[i])
    bs = [B.Batch(dm, PER_BATCH, tmax) for _ in range(2)]
    for gates in _opt.get("gates", "default").split(","):
        flags = {"default": B.RUN_EXACT_GATES, "exact": B.RUN_EXACT_GATES, "fast": B.RUN_FAST_GATES, "fast2": B.RUN_FAST_GATES2}[gates]
        t_eng0, t_eng3 = Tally("engine <-> oracle"), Tally("engine <-> oracle+blas")
        min_kmers, paired = 10 ** 9, 0
        t1 = time.time()
        for k in range(0, len(batches), 2):
            for j in (0, 1):
                bs[j].set_signals_ragged(batches[k + j])
            if PAIRED:
                bs[0].run_pair(bs[1], 1.0, flags)
            else:
                bs[0].run(1.0, flags); bs[1].run(1.0, flags)
            for j in (0, 1):
                b = bs[j]
                b.finish()
                paired += int(b.paired())
                assert b.rnn_path() == 3
                for r in range(PER_BATCH):
                    i = (k + j) * PER_BATCH + r
                    a = dict(basecall=b.basecall(r), quality=b.quality(r), psum=_path_sum(b.path(r)[0]), trans=b.transitions(r) if i < ntrans else None)
                    tag = "batch %d read %d (%d samples)" % (k + j, r, flat[i].size)
                    t_eng0.add(tag, a, ref0[i])
                    t_eng3.add(tag, a, ref3[i])
                    s = ref0[i]["basecall"]
                    min_kmers = min(min_kmers, len({s[q:q + 5] for q in range(len(s) - 4)}))
        print("\n### gates: %s%s  (engine passes %.0f s)" % (gates, {"default": " (FFHIP_RUN_EXACT_GATES: the reference's exp_ps and division, bit for bit -- the default of rounds 1-5)", "exact": " (FFHIP_RUN_EXACT_GATES: the reference's exp_ps and division, bit for bit)", "fast": " (FFHIP_RUN_FAST_GATES: hardware v_exp / v_rcp in the gate phase)", "fast2": " (FFHIP_RUN_FAST_GATES2: v_exp with a two-word exponent, v_rcp + one Newton step)"}[gates], time.time() - t1))
        print("campaign: shape %s (kind %d, H = %d, %d reads a batch%s), %d of %d batches in a paired launch; %d reads, %d samples; fewest distinct 5-mers in a read %d"
              % (SHAPE, KIND, H, PER_BATCH, ", through ffhip_batch_run_pair" if PAIRED else "", paired, len(batches), nread, sum(x.size for x in flat), min_kmers))
        for t in (t_eng0, t_eng3, t_00):
            print(t.line())
            for n in t.named[:40]:
                print("     " + n)
        both = set(n.split(":")[0] for n in t_eng0.named) & set(n.split(":")[0] for n in t_eng3.named)
        print("reads the engine calls differently from BOTH evaluations of the reference algorithm: %d%s" % (len(both), (" -- " + "; ".join(sorted(both))) if both else ""), flush=True)
    for b in bs:
        b.close()
    dm.close()


if __name__ == "__main__":
    main()
