#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/.  Run in the build container only
(needs /root/reference for part 1); the tests never read /root/reference.

  1. signal_fixtures.npz   DATA of the reference's own test fixtures src/test/raw_signal.crp,
                           trimmed_signal.crp, normalised_signal.crp (the inputs/expected outputs
                           of test_flappie_signal.c:67-111), converted from `.crp` hex-float text to
                           float32 arrays.  test_matrix.crp (5x9) likewise.
  2. oracle_net_*.npz      outputs of OUR oracle (oracle/ff_oracle.c) on seeded inputs.  These are
                           regression vectors for the oracle and the comparison target of the GPU
                           parity tests; they are NOT reference outputs (the reference's network path
                           cannot be built here -- see DESIGN.md) and say so in their `provenance` key.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/test"


def read_crp(path):
    with open(path) as fh:
        nr, nc = (int(x) for x in fh.readline().split())
        vals = [float.fromhex(tok) for tok in fh.read().split()]
    a = np.asarray(vals, dtype=np.float64).reshape(nc, nr)
    return a.astype(np.float32)


def part1():
    if not os.path.isdir(REF):
        print("reference absent; skipping part 1")
        return
    raw = read_crp(os.path.join(REF, "raw_signal.crp")).reshape(-1)
    trimmed = read_crp(os.path.join(REF, "trimmed_signal.crp")).reshape(-1)
    norm = read_crp(os.path.join(REF, "normalised_signal.crp")).reshape(-1)
    tm = read_crp(os.path.join(REF, "test_matrix.crp"))
    np.savez_compressed(os.path.join(HERE, "signal_fixtures.npz"),
                        raw=raw.astype(np.int16), trimmed=trimmed, normalised=norm, test_matrix=tm,
                        provenance="reference src/test/*.crp data (test_flappie_signal.c:21-23)")
    print("signal_fixtures.npz:", raw.shape, trimmed.shape, norm.shape, tm.shape)


def part2():
    from flappie_amd import model as M
    from oracle import ffo
    sig = np.load(os.path.join(HERE, "signal_fixtures.npz"))
    norm = sig["normalised"]
    rng = np.random.default_rng(20260928)
    # the last four: the shapes bench.py times (seed 1 = bench.py's model), short enough for the oracle and the repository
    for tag, kind, hidden, T, seed in (("lstm5_h64", M.NET_LSTM5, 64, 4000, 7), ("grumod5_h64", M.NET_GRUMOD5, 64, 2000, 7),
                                       ("lstm5_h96_t1237", M.NET_LSTM5, 96, 1237, 7),
                                       ("lstm5_h256_t1500", M.NET_LSTM5, 256, 1500, 1), ("lstm5_h384_t1500", M.NET_LSTM5, 384, 1500, 1),
                                       ("lstm5_h512_t1000", M.NET_LSTM5, 512, 1000, 1), ("grumod5_h256_t1000", M.NET_GRUMOD5, 256, 1000, 1)):
        mdl = M.synthetic_model(kind, hidden, seed=seed)
        om = ffo.OracleModel(mdl)
        reads = [norm[1000:1000 + T].copy(), rng.standard_normal(T).astype(np.float32)]
        out = {}
        for i, r in enumerate(reads):
            res = om.basecall(r)
            resv = om.basecall(r, viterbi_only=True, want_trans=False)
            out["signal%d" % i] = r
            out["trans%d" % i] = res["trans"]
            out["post%d" % i] = res["post"]
            out["path%d" % i] = res["path"]
            out["qpath%d" % i] = res["qpath"]
            out["trace%d" % i] = res["trace"].astype(np.uint8)
            out["basecall%d" % i] = np.frombuffer(res["basecall"].encode(), dtype=np.uint8)
            out["quality%d" % i] = np.frombuffer(res["quality"].encode(), dtype=np.uint8)
            out["score%d" % i] = np.float32(res["score"])
            out["vit_basecall%d" % i] = np.frombuffer(resv["basecall"].encode(), dtype=np.uint8)
            out["vit_path%d" % i] = resv["path"]
        np.savez_compressed(os.path.join(HERE, "oracle_net_%s.npz" % tag), kind=kind, hidden=hidden, seed=seed,
                            provenance="oracle/ff_oracle.c output (NOT reference output; parity unpinned)",
                            **out)
        kmers = [len({bytes(out["basecall%d" % i][k:k + 5]) for k in range(out["basecall%d" % i].size - 4)}) for i in (0, 1)]
        print(tag, "nblock", out["path0"].size - 1, "basecall len", out["basecall0"].size, out["basecall1"].size, "distinct 5-mers", kmers)


if __name__ == "__main__":
    part1()
    part2()
