#!/usr/bin/env python3
"""Config 5 against the oracle AT ITS REAL LENGTH: reads of 100 000 samples -- 20 000 dependent steps per layer at the
r103 shape (LSTM5, H = 512, bench.py's c5 model, seed 1), 50 000 blocks and 10 states for one read at the r941_5mC shape (GRUmod5, H = 256,
stride 2, bench.py's c4 model).  What only shows at length: drift of the split products over 5 x 20 000 steps, the fp64 linear-space CRF
chains' power-of-two rescaling over 20 000 blocks (k_crf_fb), the traceback's 2048-block LDS chunks, k_trace at 20 001 columns.
Reference path: networks.c:539-586, decode.c:119-204, 377-543.

The oracle costs ~165 s per such read per core, so its half runs where CPU time is free and its output travels as a scratch file:

  tools/parity_long.py oracle OUT.npz [nlstm=16] [ngru=1]      CPU (process pool): seeded reads -> oracle calls, paths, scores, posteriors, traces
  tools/parity_long.py gpu IN.npz                              GPU box: the same reads through the engine (batches of 16, trace on), the comparison
  tools/parity_long.py golden IN.npz OUT.npz [nkeep=4]         the slice the test suite carries (tests/golden/long_reads.npz): calls, qualities, paths,
                                                               trace columns and every 16th block of trans / post of the first nkeep LSTM reads + the GRUmod read
Read r is default_rng(5000 + r).standard_normal(100000) as float32 (LSTM reads 0 .. nlstm-1, GRUmod reads 100 ..)."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

T = 100000
SHAPES = {"lstm": (M.NET_LSTM5, 512), "gru": (M.NET_GRUMOD5, 256)}
_om = {}


def signal(tag, r):
    return np.random.default_rng(5000 + (r if tag == "lstm" else 100 + r)).standard_normal(T).astype(np.float32)


def _call(job):
    tag, r = job
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle import ffo
    if tag not in _om:
        kind, hidden = SHAPES[tag]
        _om[tag] = ffo.OracleModel(M.synthetic_model(kind, hidden, seed=1))
    t0 = time.time()
    o = _om[tag].basecall(signal(tag, r))
    return tag, r, dict(basecall=o["basecall"], quality=o["quality"], path=np.asarray(o["path"], np.int8), trans=np.asarray(o["trans"], np.float32),
                        post=np.asarray(o["post"], np.float32), trace=np.asarray(o["trace"]).astype(np.uint8), score=float(o["score"])), time.time() - t0


def cmd_oracle(out, nlstm, ngru):
    jobs = [("lstm", r) for r in range(nlstm)] + [("gru", r) for r in range(ngru)]
    ncpu = len(os.sched_getaffinity(0))
    t0 = time.time()
    store = {}
    with mp.Pool(min(ncpu, len(jobs))) as pool:
        for tag, r, o, dt in pool.imap_unordered(_call, jobs):
            print("oracle %s read %d: %d bases, %.0f s (%.0f s since start)" % (tag, r, len(o["basecall"]), dt, time.time() - t0), flush=True)
            for k, v in o.items():
                store["%s_%d_%s" % (tag, r, k)] = np.asarray(v)
    store["nlstm"], store["ngru"] = np.int64(nlstm), np.int64(ngru)
    np.savez(out, **store)
    print("wrote %s (%d + %d reads, %.0f s on %d CPUs)" % (out, nlstm, ngru, time.time() - t0, ncpu))


def compare(tag, r, ref, b, slot, lines, tally):
    """engine results of batch slot `slot` against the oracle's record `ref` of (tag, r); ref holds full arrays or the golden slice (every `sub`-th block)"""
    sub = int(ref.get("sub", 1))
    trans, post = b.transitions(slot), b.posterior(slot)
    nblk = trans.shape[0]
    tsel = trans[::sub] if sub > 1 else trans
    psel = post[::sub] if sub > 1 else post
    assert tsel.shape == ref["trans"].shape, (tsel.shape, ref["trans"].shape)
    dt = np.abs(tsel - ref["trans"])
    worst = float(dt.max())
    at = int(np.unravel_index(int(dt.argmax()), dt.shape)[0]) * sub
    fin = np.isfinite(ref["post"]) & np.isfinite(psel)
    dp = float(np.abs(psel[fin] - ref["post"][fin]).max()) if fin.any() else 0.0
    path = np.asarray(b.path(slot)[0])
    path_diff = int((path.astype(np.int64) != ref["path"].astype(np.int64)).sum())
    first = int(np.argmax(path.astype(np.int64) != ref["path"].astype(np.int64))) if path_diff else -1
    bases_equal = b.basecall(slot) == str(ref["basecall"])
    qual_equal = b.quality(slot) == str(ref["quality"])
    tr = np.asarray(b.trace(slot)).astype(np.int64)
    rt = ref["trace"].astype(np.int64)
    if tr.shape != rt.shape:
        tr = tr.T
    assert tr.shape == rt.shape, (tr.shape, rt.shape)
    tdiff = np.abs(tr - rt)
    tally["reads"] += 1
    tally["bases"] += len(str(ref["basecall"]))
    tally["worst_trans"] = max(tally["worst_trans"], worst)
    tally["worst_post"] = max(tally["worst_post"], dp)
    tally["base_mismatch"] += 0 if bases_equal else 1
    tally["qual_mismatch"] += 0 if (qual_equal or not bases_equal) else 1
    tally["path_mismatch"] += 1 if path_diff else 0
    tally["trace_cells"] += tdiff.size
    tally["trace_off1"] += int((tdiff == 1).sum())
    tally["trace_more"] += int((tdiff > 1).sum())
    lines.append("%s read %d: %d blocks, %d bases; worst |dtrans| %.3e at block %d of %d (%.0f %% along the read), worst |dlogpost| %.3e; path %s; bases %s; quality %s; "
                 "trace cells off by one %d of %d, by more %d"
                 % (tag, r, nblk, len(str(ref["basecall"])), worst, at, nblk, 100.0 * at / max(1, nblk), dp,
                    "equal" if not path_diff else "%d of %d blocks differ, first at block %d" % (path_diff, path.size, first),
                    "equal" if bases_equal else "DIFFER (%d against %d)" % (len(b.basecall(slot)), len(str(ref["basecall"]))), "equal" if qual_equal else "differ",
                    int((tdiff == 1).sum()), tdiff.size, int((tdiff > 1).sum())))
    return worst, dp, path_diff, bases_equal


def load(path):
    z = np.load(path, allow_pickle=False)
    recs = {}
    for key in z.files:
        if key in ("nlstm", "ngru", "sub"):
            continue
        tag, r, field = key.split("_", 2)
        recs.setdefault((tag, int(r)), {})[field] = z[key]
    if "sub" in z.files:
        for rec in recs.values():
            rec["sub"] = int(z["sub"])
    return recs


def run_engine(recs, per_batch=16):
    """every record's read through the engine (trace on), compared: -> report lines, tallies by tag"""
    from flappie_amd import binding as B
    eng = B.Engine(0)
    lines, tallies = [], {}
    for tag in ("lstm", "gru"):
        rs = sorted(r for (t, r) in recs if t == tag)
        if not rs:
            continue
        kind, hidden = SHAPES[tag]
        dm = B.DeviceModel(eng, M.synthetic_model(kind, hidden, seed=1))
        tally = tallies.setdefault(tag, dict(reads=0, bases=0, worst_trans=0.0, worst_post=0.0, base_mismatch=0, qual_mismatch=0, path_mismatch=0, trace_cells=0,
                                             trace_off1=0, trace_more=0, rnn_path=None))
        for k in range(0, len(rs), per_batch):
            part = rs[k:k + per_batch]
            b = B.Batch(dm, len(part), T)
            b.set_signals(np.stack([signal(tag, r) for r in part]))
            b.run(); b.finish()
            tally["rnn_path"] = b.rnn_path()
            for slot, r in enumerate(part):
                compare(tag, r, recs[(tag, r)], b, slot, lines, tally)
            b.close()
        dm.close()
    eng.close()
    return lines, tallies


def cmd_gpu(inp):
    t0 = time.time()
    lines, tallies = run_engine(load(inp))
    print("config 5 at its real length: reads of %d samples, engine (default path, trace on) against the oracle (dot mode 0); %.0f s" % (T, time.time() - t0))
    for ln in lines:
        print("  " + ln)
    for tag, t in tallies.items():
        kind, hidden = SHAPES[tag]
        print("%s (kind %d, H = %d, recurrent kernel path %s): %d reads, %d bases: worst |dtrans| %.3e (north_star 1e-4), worst |dlogpost| %.3e; %d reads with another base "
              "string, %d more with another quality string, %d with another Viterbi path; trace: %d of %d cells off by one (%.4f %%), %d by more"
              % (tag, kind, hidden, t["rnn_path"], t["reads"], t["bases"], t["worst_trans"], t["worst_post"], t["base_mismatch"], t["qual_mismatch"], t["path_mismatch"],
                 t["trace_off1"], t["trace_cells"], 100.0 * t["trace_off1"] / max(1, t["trace_cells"]), t["trace_more"]))


def cmd_golden(inp, out, nkeep, sub=16):
    z = np.load(inp, allow_pickle=False)
    keep = {}
    for key in z.files:
        if key in ("nlstm", "ngru"):
            continue
        tag, r, field = key.split("_", 2)
        if tag == "lstm" and int(r) >= nkeep:
            continue
        v = z[key]
        if field in ("trans", "post"):
            v = v[::sub]
        keep[key] = v
    keep["sub"] = np.int64(sub)
    np.savez_compressed(out, **keep)
    print("wrote %s: %d arrays, %.1f MB" % (out, len(keep), os.path.getsize(out) / 1e6))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "oracle":
        cmd_oracle(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 16, int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    elif len(sys.argv) >= 3 and sys.argv[1] == "gpu":
        cmd_gpu(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "golden":
        cmd_golden(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 4)
    else:
        sys.exit(__doc__)
