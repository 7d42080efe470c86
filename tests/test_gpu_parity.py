"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs, against the committed vectors, and -- at BASELINE.json's full size -- through
size-independent properties.

Tolerances: BASELINE.json north_star asks for transition scores within 1e-4 (absolute, fp32) and a bit-exact called base
string; the suite holds the scores to 5e-5 (measured worst: 2.1e-5).  Integer outputs (path, trace) are compared exactly except `trace`, whose
round(255 p) may differ by one count where p sits on a rounding boundary."""
import os

import numpy as np
import pytest

from flappie_amd import model as M

pytestmark = pytest.mark.gpu

# Round 4: on the input-driven, well-conditioned models (flappie_amd/model.py SYNTH_GAINS) the whole GPU suite sees worst |dtrans| 2.1e-5 and worst
# end-to-end |dlogpost| 4.0e-5 over 434 reads (printed at the end of every run: tests/conftest.py) -- the bounds are HALF north_star's now, not at it
TOL_SCORE = 5e-5
TOL_POST = 1e-4          # log posterior END TO END (see compare_read); the kernel alone: 2e-5 + 2e-6 |x| (tests/test_decode_gpu.py)
_trace_cells = [0, 0]    # [cells compared, cells off by one count] of the current test


@pytest.fixture(autouse=True)
def _report_trace_rate():
    _trace_cells[0] = _trace_cells[1] = 0
    yield
    if _trace_cells[0]:
        print("trace: %d of %d cells differ from the oracle by one count (%.4f %%)" % (_trace_cells[1], _trace_cells[0], 100.0 * _trace_cells[1] / _trace_cells[0]))


@pytest.fixture(scope="module")
def B():
    from flappie_amd import binding
    return binding


@pytest.fixture(scope="module")
def ffo():
    from oracle import ffo as _ffo
    return _ffo


def run_batch(B, engine, mdl, sig, flags=0, temperature=1.0):
    dm = B.DeviceModel(engine, mdl)
    b = B.Batch(dm, sig.shape[0], sig.shape[1])
    b.set_signals(sig)
    b.run(temperature, flags)
    b.finish()
    return dm, b


def compare_read(b, r, ref, viterbi=False):
    tr = b.transitions(r)
    assert np.isfinite(tr).all()
    from conftest import note_parity
    note_parity(np.abs(tr - ref["trans"]).max(), None if viterbi else np.abs(b.posterior(r) - ref["post"]).max())
    assert np.abs(tr - ref["trans"]).max() <= TOL_SCORE
    path, qpath = b.path(r)
    assert np.array_equal(path, ref["path"])
    assert np.isnan(qpath[0])
    assert np.abs(qpath[1:] - ref["qpath"][1:]).max() <= TOL_POST          # (the path's per-block scores ARE posterior entries unless viterbi-only)
    assert b.basecall(r) == ref["basecall"]
    assert b.quality(r) == ref["quality"]
    assert abs(b.score(r) - ref["score"]) <= 2e-3 * max(1.0, abs(ref["score"]) * 1e-2)
    if not viterbi:
        # end to end: the scores' own deviation (<= 1e-4) propagates through two log-sum-exp recursions -- measured up to 1.2e-4 at |dtrans| = 2.6e-5 --
        # so this bound cannot be the posterior kernel's; THAT is held to 2e-5 + 2e-6 |x| on identical scores in tests/test_decode_gpu.py
        assert np.abs(b.posterior(r) - ref["post"]).max() <= TOL_POST
        dt = np.abs(b.trace(r) - ref["trace"])
        assert dt.max() <= 1          # round(255 p) at a rounding boundary; the rate is printed with every test (-rP / -s)
        _trace_cells[0] += dt.size
        _trace_cells[1] += int((dt == 1).sum())


CASES = [
    # kind, hidden, T, nread
    (M.NET_LSTM5, 64, 4000, 3),       # T % 5 == 0: the reference's right-edge column shift fires
    (M.NET_LSTM5, 64, 4003, 2),       # T % 5 != 0
    (M.NET_LSTM5, 96, 1237, 17),      # ragged: more than one read tile, partially filled
    (M.NET_LSTM5, 36, 601, 2),        # hidden size not a multiple of 16 (zero-padded units)
    (M.NET_GRUMOD5, 64, 2000, 3),     # modified-base model family: GRU, stride 2 (even T quirk), ACGTZ
    (M.NET_GRUMOD5, 48, 1501, 2),
]


@pytest.mark.parametrize("kind,hidden,T,nread", CASES)
def test_basecall_matches_oracle(B, ffo, engine, kind, hidden, T, nread):
    mdl = M.synthetic_model(kind, hidden, seed=7)
    om = ffo.OracleModel(mdl)
    sig = np.random.default_rng(1000 + T).standard_normal((nread, T)).astype(np.float32)
    dm, b = run_batch(B, engine, mdl, sig)
    try:
        for r in range(nread):
            compare_read(b, r, om.basecall(sig[r]))
    finally:
        b.close(); dm.close()


def test_viterbi_only_and_temperature(B, ffo, engine):
    mdl = M.synthetic_model(M.NET_LSTM5, 64, seed=9)
    om = ffo.OracleModel(mdl)
    sig = np.random.default_rng(5).standard_normal((2, 1500)).astype(np.float32)
    dm, b = run_batch(B, engine, mdl, sig, flags=B.RUN_VITERBI_ONLY, temperature=0.7)
    try:
        for r in range(2):
            compare_read(b, r, om.basecall(sig[r], temperature=0.7, viterbi_only=True), viterbi=True)
        with pytest.raises(B.FFHipError):
            b.posterior(0)
    finally:
        b.close(); dm.close()


def test_layer_by_layer_activations(B, ffo, engine):
    """Every stage against the oracle's stage, so that a whole-network tolerance cannot hide a wrong layer."""
    import ctypes as C
    mdl = M.synthetic_model(M.NET_LSTM5, 64, seed=11)
    sig = np.random.default_rng(6).standard_normal((1, 2000)).astype(np.float32)
    dm, b = run_batch(B, engine, mdl, sig, flags=B.RUN_KEEP_ACTS)
    try:
        L = ffo.lib()
        x = ffo.HostMat.from_dense(sig[0].reshape(-1, 1))
        cur = x.ptr
        for cv in mdl.convs:
            nxt = L.fo_convolution(cur, ffo.HostMat.from_model_mat(cv.W).ptr, ffo.HostMat.from_model_mat(cv.b).ptr, cv.stride)
            L.fo_swish_inplace(nxt)
            cur = nxt
        ref = ffo.take(cur, free=False)
        assert np.abs(b.activation(-1, 0) - ref).max() <= 1e-5
        for l, r in enumerate(mdl.rnns):
            xa = L.fo_affine_map(cur, ffo.HostMat.from_model_mat(r.iW).ptr, ffo.HostMat.from_model_mat(r.b).ptr)
            cur = L.fo_lstm(xa, ffo.HostMat.from_model_mat(r.sW).ptr, int(l % 2 == 0))
            ref = ffo.take(cur, free=False)
            assert np.abs(b.activation(l, 0) - ref).max() <= 5e-5, "layer %d" % l
    finally:
        b.close(); dm.close()


@pytest.mark.parametrize("tag", ["lstm5_h64", "grumod5_h64", "lstm5_h96_t1237", "lstm5_h256_t1500", "lstm5_h384_t1500", "lstm5_h512_t1000", "grumod5_h256_t1000"])
def test_against_committed_vectors(B, engine, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "oracle_net_%s.npz" % tag))
    mdl = M.synthetic_model(int(g["kind"]), int(g["hidden"]), seed=int(g["seed"]))
    sig = np.stack([g["signal0"], g["signal1"]])
    dm, b = run_batch(B, engine, mdl, sig)
    try:
        for i in (0, 1):
            assert np.abs(b.transitions(i) - g["trans%d" % i]).max() <= TOL_SCORE
            assert b.basecall(i).encode() == g["basecall%d" % i].tobytes()
            assert b.quality(i).encode() == g["quality%d" % i].tobytes()
            assert np.array_equal(b.path(i)[0], g["path%d" % i])
            assert np.abs(b.trace(i) - g["trace%d" % i].astype(np.int32)).max() <= 1
    finally:
        b.close(); dm.close()


@pytest.mark.parametrize("kind,hidden,nread,T", [(M.NET_LSTM5, 96, 18, 1500), (M.NET_GRUMOD5, 64, 18, 1500), (M.NET_LSTM5, 384, 40, 600),
                                                 (M.NET_GRUMOD5, 256, 40, 400), (M.NET_LSTM5, 512, 33, 500)])
def test_recurrent_kernel_variants_agree(B, engine, kind, hidden, nread, T):
    """Three implementations of the recurrent stack must agree to rounding: the fused persistent
    layer (projection + recurrence in one launch), the persistent recurrence behind a separate
    projection GEMM, and the launch-per-step kernels.  Each is also the others' cross-check on
    shapes the oracle is too slow for."""
    mdl = M.synthetic_model(kind, hidden, seed=13)
    sig = np.random.default_rng(77).standard_normal((nread, T)).astype(np.float32)       # 2 or 3 read tiles, the last one partly filled
    outs = []
    for flags in (0, B.RUN_UNFUSED_RNN, B.RUN_STEPWISE_RNN):
        dm, b = run_batch(B, engine, mdl, sig, flags=flags)
        outs.append(([b.transitions(r) for r in (0, nread // 2, nread - 1)], [b.basecall(r) for r in range(nread)]))
        b.close(); dm.close()
    for tr, calls in outs[1:]:
        for a, c in zip(outs[0][0], tr):
            assert np.abs(a - c).max() <= 5e-5      # fp32 summation order differs between the variants
        assert calls == outs[0][1]


def test_raw_table_entry_point_and_batch_independence(B, ffo, engine):
    """set_reads (raw_table, start/end honoured) == set_signals; a read's result does not depend on
    which other reads share its batch (reads are independent units, flappie.c:364-385)."""
    mdl = M.synthetic_model(M.NET_LSTM5, 64, seed=7)
    rng = np.random.default_rng(8)
    T = 1000
    raws = [rng.standard_normal(T + 300).astype(np.float32) for _ in range(3)]
    starts = [0, 123, 300]
    sig = np.stack([r[s:s + T] for r, s in zip(raws, starts)])
    dm = B.DeviceModel(engine, mdl)
    b1 = B.Batch(dm, 3, T)
    b1.set_reads(raws, starts)
    b1.run(); b1.finish()
    b2 = B.Batch(dm, 1, T)
    b2.set_signals(sig[2:3])
    b2.run(); b2.finish()
    try:
        assert np.array_equal(b1.transitions(2), b2.transitions(0))
        assert b1.basecall(2) == b2.basecall(0)
        om = ffo.OracleModel(mdl)
        compare_read(b1, 1, om.basecall(sig[1]))
    finally:
        b1.close(); b2.close(); dm.close()


def test_error_paths(B, engine):
    mdl = M.synthetic_model(M.NET_LSTM5, 32, seed=1)
    dm = B.DeviceModel(engine, mdl)
    try:
        with pytest.raises(B.FFHipError):
            B.Batch(dm, 1, 10)            # shorter than the convolution window: outside the reference's domain
        with pytest.raises(B.FFHipError):
            B.Batch(dm, 0, 1000)          # empty batch
        b = B.Batch(dm, 1, 1000)
        with pytest.raises(B.FFHipError):
            b.finish()                    # run() not called
        b.close()
    finally:
        dm.close()


def test_full_size_properties(B, engine):
    """BASELINE.json config 2 (r941_native shape, 256 reads x 4000 samples).  The oracle needs seconds
    per read at this size, so parity is checked through size-independent properties:
      * duplicated reads give identical results wherever they sit in the batch;
      * the transition scores are globally normalised: the CRF partition function of the
        returned matrix is zero per block (logZ/nblock was subtracted, layers.c:1089-1096);
      * log-posteriors normalise to one per block; Viterbi qpath sums to the path score;
      * a handful of reads are checked against the oracle itself."""
    from oracle import ffo
    H, nread, T = 384, 256, 4000
    mdl = M.synthetic_model(M.NET_LSTM5, H, seed=1)
    rng = np.random.default_rng(20260928)
    sig = rng.standard_normal((nread, T)).astype(np.float32)
    sig[200] = sig[3]
    sig[255] = sig[17]
    dm, b = run_batch(B, engine, mdl, sig)
    try:
        assert b.nblock == 800
        assert np.array_equal(b.transitions(200), b.transitions(3))
        assert b.basecall(255) == b.basecall(17) and b.quality(255) == b.quality(17)
        for r in (0, 100, 255):
            tr = b.transitions(r)
            logZ = ffo.lib().fo_partition_function(ffo.HostMat.from_dense(tr).ptr)
            assert abs(logZ) / 800 <= 2e-6
            post = b.posterior(r)
            assert np.abs(np.exp(post.astype(np.float64)).sum(axis=1) - 1.0).max() <= 1e-4
            path, qpath = b.path(r)
            assert abs(float(qpath[1:].astype(np.float64).sum()) - b.score(r)) <= 1e-2
            assert path.min() >= 0 and path.max() < 8
            trc = b.trace(r)
            assert trc.min() >= 0 and trc.max() <= 255
            assert np.abs(trc[1:].sum(axis=1) - 255).max() <= 4
        om = ffo.OracleModel(mdl)
        for r in (0, 131):
            compare_read(b, r, om.basecall(sig[r]))
    finally:
        b.close(); dm.close()


@pytest.mark.parametrize("label,kind,H,nread,T,nstate", [
    ("config 4: r941_5mC shape (GRUmod, stride 2, 10 states)", M.NET_GRUMOD5, 256, 48, 4000, 10),
    ("config 5: r103 shape, 100 000-sample reads with trace", M.NET_LSTM5, 512, 16, 100000, 8),
])
def test_other_baseline_configs_properties(B, engine, label, kind, H, nread, T, nstate):
    """BASELINE.json configs 4 and 5 at their real shapes, through the size-independent properties (the oracle
    needs minutes per read here): duplicates identical, global normalisation, posteriors sum to one, trace bounded."""
    from oracle import ffo
    mdl = M.synthetic_model(kind, H, seed=2)
    rng = np.random.default_rng(H + T)
    sig = rng.standard_normal((nread, T)).astype(np.float32)
    sig[nread - 1] = sig[1]
    dm, b = run_batch(B, engine, mdl, sig)
    try:
        nblock = b.nblock
        assert nblock == mdl.nblock(T)
        assert np.array_equal(b.transitions(nread - 1), b.transitions(1))
        assert b.basecall(nread - 1) == b.basecall(1) and b.quality(nread - 1) == b.quality(1)
        for r in (0, nread // 2):
            tr = b.transitions(r)
            assert np.isfinite(tr).all()
            logZ = ffo.lib().fo_partition_function(ffo.HostMat.from_dense(tr).ptr)
            assert abs(logZ) / nblock <= 2e-6, label
            post = b.posterior(r)
            assert np.abs(np.exp(post.astype(np.float64)).sum(axis=1) - 1.0).max() <= 2e-4
            path, qpath = b.path(r)
            assert path.min() >= 0 and path.max() < nstate
            trc = b.trace(r)
            assert trc.shape == (nblock + 1, nstate) and trc.min() >= 0 and trc.max() <= 255
            assert len(b.basecall(r)) == len(b.quality(r)) > 0
    finally:
        b.close(); dm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,hidden", [(M.NET_GRUMOD5, 64), (M.NET_LSTM5, 64)])
def test_decode_kernels_agree_with_their_chain_order_forms(B, engine, kind, hidden):
    """The butterfly decode kernels (k_transpost8/10, k_viterbi8/10) against the chain-order / generic forms selected by
    FFHIP_DEBUG=exact_order: the same Viterbi paths, qualities and calls; posteriors equal up to the rounding of the summation order."""
    mdl = M.synthetic_model(kind, hidden, seed=21)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(77)
    lens = [2500, 19, 777, 0, 1234, 2499, 300, 1500, 64, 2000]
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    out = {}
    for mode in ("fast", "exact"):
        if mode == "exact":
            os.environ["FFHIP_DEBUG"] = "exact_order"
        try:
            b = B.Batch(dm, len(sigs), max(lens))
            b.set_signals_ragged(sigs)
            b.run()
            b.finish()
            out[mode] = [(b.basecall(r), b.quality(r), b.path(r), b.posterior(r), b.score(r)) if lens[r] else None for r in range(len(sigs))]
            b.close()
        finally:
            os.environ.pop("FFHIP_DEBUG", None)
    for r, n in enumerate(lens):
        if not n:
            continue
        f, e = out["fast"][r], out["exact"][r]
        assert f[0] == e[0] and f[1] == e[1], r
        assert np.array_equal(f[2][0], e[2][0]), r                                   # Viterbi path
        assert np.abs(f[2][1][1:] - e[2][1][1:]).max() <= 2e-5                       # its per-block scores are posteriors
        # log posteriors: improbable transitions sit at -100 and below, where one ulp is 1e-5
        assert np.all(np.abs(f[3] - e[3]) <= 2e-5 + 2e-6 * np.abs(e[3])), (r, float(np.abs(f[3] - e[3]).max()))
        assert abs(f[4] - e[4]) <= 1e-3 * max(1.0, abs(e[4]))
    dm.close()
