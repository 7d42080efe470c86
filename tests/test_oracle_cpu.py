"""CPU tests of the oracle (oracle/ff_oracle.c): pinned against the reference's own fixtures and
known-answer values, against the part of the reference that builds here (oracle/_ref), against an
independent numpy restatement, and against the committed regression vectors."""
import ctypes as C
import os

import numpy as np
import pytest

from flappie_amd import model as M
from oracle import ffo

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def sig():
    return np.load(os.path.join(HERE, "golden", "signal_fixtures.npz"))


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---- reference fixtures: test_flappie_signal.c:67-111 -------------------------------------------
def _scaled_raw(sig):
    # test_flappie_signal.c:74-82: range 1373.41, digitisation 8192, offset 16
    unit = np.float32(1373.41) / np.float32(8192.0)
    return ((sig["raw"].astype(np.float32) + np.float32(16.0)) * unit).astype(np.float32)


def test_trim_signal_fixture(sig):
    raw = _scaled_raw(sig)
    start, end = C.c_size_t(0), C.c_size_t(raw.size)
    assert ffo.lib().fo_trim_raw_by_mad(_f(raw), C.byref(start), C.byref(end), 100, 0.0) == 0
    assert start.value == 0
    assert end.value == (raw.size // 100) * 100
    trimmed = raw[start.value + 200: end.value - 10]
    assert trimmed.size == sig["trimmed"].size
    assert np.abs(trimmed - sig["trimmed"]).max() <= 1e-4          # tolerance of the reference test


def test_normalise_signal_fixture(sig):
    x = sig["trimmed"].copy()
    ffo.lib().fo_medmad_normalise_array(_f(x), x.size)
    assert np.abs(x - sig["normalised"]).max() <= 1e-5             # tolerance of the reference test


# ---- the reference itself, where it builds (util.c + flappie_common.c -> oracle/_ref) -----------
def _sigref():
    path = os.path.join(ROOT, "oracle", "_ref", "libflappie_sigref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference sources absent on this machine)")
    L = C.CDLL(path)
    L.medianf.restype = C.c_float
    L.medianf.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    L.madf.restype = C.c_float
    L.madf.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float)]
    L.medmad_normalise_array.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    L.quantilef.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t]
    L.argmaxf.restype = C.c_int
    L.argmaxf.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    return L


class RawTable(C.Structure):
    _fields_ = [("uuid", C.c_char_p), ("n", C.c_size_t), ("start", C.c_size_t), ("end", C.c_size_t),
                ("raw", C.POINTER(C.c_float))]


def test_signal_prep_matches_compiled_reference(sig):
    L = _sigref()
    rng = np.random.default_rng(5)
    for n in (2, 3, 10, 101, 4000):
        x = rng.standard_normal(n).astype(np.float32)
        assert L.medianf(_f(x), n) == ffo.lib().fo_medianf(_f(x), n)
        assert L.madf(_f(x), n, None) == ffo.lib().fo_madf(_f(x), n, None)
        a, b = x.copy(), x.copy()
        L.medmad_normalise_array(_f(a), n)
        ffo.lib().fo_medmad_normalise_array(_f(b), n)
        assert np.array_equal(a, b)
        for q in (0.0, 0.25, 0.5, 0.9, 1.0):
            p1, p2 = C.c_float(q), C.c_float(q)
            L.quantilef(_f(x), n, C.byref(p1), 1)
            ffo.lib().fo_quantilef(_f(x), n, C.byref(p2), 1)
            assert p1.value == p2.value
    # trim_and_segment_raw on the reference's raw fixture (passed and returned by value)
    raw = _scaled_raw(sig)
    L.trim_and_segment_raw.restype = RawTable
    L.trim_and_segment_raw.argtypes = [RawTable, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
    for thresh in (0.0, 0.3, 0.5):
        rt = RawTable(None, raw.size, 0, raw.size, _f(raw))
        out = L.trim_and_segment_raw(rt, 200, 10, 100, thresh)
        s, e = C.c_size_t(0), C.c_size_t(raw.size)
        rc = ffo.lib().fo_trim_and_segment_raw(_f(raw), raw.size, C.byref(s), C.byref(e), 200, 10, 100, thresh)
        assert rc == 0
        assert (out.start, out.end) == (s.value, e.value)


# ---- known-answer values of the reference's unit tests -----------------------------------------
def test_elu_known_answers():
    # test_flappie_elu.c:31-79
    L = ffo.lib()
    assert L.fo_eluf(0.0) == 0.0 and L.fo_eluf(-0.0) == 0.0
    for v in (1.0, 2.0, 3.0, 4.0):
        assert L.fo_eluf(v) == v
    for v, want in ((-1.0, -0.6321206), (-2.0, -0.8646647), (-3.0, -0.9502129), (-4.0, -0.9816844)):
        assert abs(L.fo_eluf(v) - want) <= 1e-6


def test_median_known_answers():
    # test_util.c:32-42
    a = np.array([0, 1, 2, 3, 4], dtype=np.float32)
    assert ffo.lib().fo_medianf(_f(a), 5) == 2.0
    assert abs(ffo.lib().fo_medianf(_f(a), 4) - 1.5) <= 1e-5


@pytest.mark.parametrize("nr", [8, 9, 10, 11])
def test_row_normalise_pad_masking(nr):
    # test_flappie_matrix.c:32-58: every stored element (pads included) is 1 before the call
    stride = 4 * ((nr + 3) // 4)
    m = ffo.HostMat(nr, 1, np.ones((1, stride), dtype=np.float32))
    ffo.lib().fo_row_normalise_inplace(m.ptr)
    assert np.abs(m.data[0, :nr] - 1.0 / nr).max() <= 1e-5


def test_identity_convolution():
    # test_flappie_convolution.c:395-416: 1-tap identity filter, stride 1, odd and even lengths
    for n in (9, 10, 37):
        x = np.arange(1, n + 1, dtype=np.float32).reshape(n, 1)
        X = ffo.HostMat.from_dense(x)
        W = ffo.HostMat(1, 1, np.array([[1, 0, 0, 0]], dtype=np.float32))
        b = ffo.HostMat(1, 1, np.zeros((1, 4), dtype=np.float32))
        out = ffo.take(ffo.lib().fo_convolution(X.ptr, W.ptr, b.ptr, 1))
        assert np.abs(out[:, 0] - x[:, 0]).max() <= 1e-5


# ---- convolution: independent numpy restatement of SURVEY.md section 8a row A3 --------------------
def numpy_conv_recipe(x, taps, bias, stride):
    """x[T, nf]; taps[nfilter, winlen, nf].  float64 evaluation of the reference's three regions."""
    T, nf = x.shape
    nfilter, winlen, _ = taps.shape
    s = stride
    padL, padR = (winlen - 1) // 2, winlen // 2
    Tout = -(-T // s)
    ncolsL = -(-padL // s)
    shiftX = ncolsL * s - padL
    nstepC = -(-winlen // s)
    nstepX = s * nstepC
    xp = np.zeros((T + 2 * winlen, nf))
    xp[winlen:winlen + T] = x
    out = np.tile(bias.astype(np.float64), (Tout, 1))

    def add(col, x0):
        if 0 <= col < Tout:
            win = xp[x0 + winlen: x0 + winlen + winlen]            # zero outside [0, T)
            out[col] += np.einsum("fwj,wj->f", taps.astype(np.float64), win)
    for w in range(0, padL, s):
        add(w // s, w - padL)
    for w in range(0, winlen, s):
        for k in range((T - shiftX - w) // nstepX):
            add(ncolsL + w // s + nstepC * k, shiftX + w + nstepX * k)
    maxCol, rem = (T - shiftX) // nstepX, (T - shiftX) % nstepX
    colR = ncolsL + nstepC * (maxCol - 1) + rem // s + 1
    startR = s - (padL + T - winlen) % s - 1
    for w in range(startR, padR, s):
        add(colR + w // s, T - winlen + 1 + w)
    return out


def naive_conv(x, taps, bias, stride):
    T, nf = x.shape
    nfilter, winlen, _ = taps.shape
    padL = (winlen - 1) // 2
    Tout = -(-T // stride)
    xp = np.zeros((T + 2 * winlen, nf))
    xp[winlen:winlen + T] = x
    out = np.tile(bias.astype(np.float64), (Tout, 1))
    for c in range(Tout):
        x0 = c * stride - padL
        out[c] += np.einsum("fwj,wj->f", taps.astype(np.float64), xp[x0 + winlen: x0 + 2 * winlen])
    return out


CONV_CASES = [(1, 4, 5, 1), (4, 16, 5, 1), (16, 24, 19, 5), (1, 8, 19, 2), (16, 8, 19, 3), (16, 8, 7, 5), (3, 5, 11, 2)]


@pytest.mark.parametrize("nf,nfilter,winlen,stride", CONV_CASES)
@pytest.mark.parametrize("T", [23, 100, 105, 399, 400, 401, 402, 403, 404, 405])
def test_convolution_against_numpy_recipe(nf, nfilter, winlen, stride, T):
    rng = np.random.default_rng(T * 131 + winlen)
    cm = M._conv_mat(rng, nf, nfilter, winlen)
    bias = rng.uniform(-0.1, 0.1, nfilter).astype(np.float32)
    layer = M.ConvLayer(cm, M.Mat.vector(bias), stride, nf, winlen)
    x = rng.standard_normal((T, nf)).astype(np.float32)
    X = ffo.HostMat.from_dense(x)
    got = ffo.take(ffo.lib().fo_convolution(X.ptr, ffo.HostMat.from_model_mat(cm).ptr,
                                            ffo.HostMat.from_model_mat(layer.b).ptr, stride))
    want = numpy_conv_recipe(x, layer.taps(), bias, stride)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5
    if stride == 1:
        # stride-1 convolutions equal the naive zero-padded definition exactly
        assert np.abs(got - naive_conv(x, layer.taps(), bias, 1)).max() <= 2e-5


def test_stride5_right_edge_quirk_T4000():
    """T % 5 == 0: column Tb-2 holds the naive value of column Tb-1, column Tb-1 is bias only."""
    rng = np.random.default_rng(11)
    nf, nfilter, winlen, stride, T = 16, 8, 19, 5, 4000
    cm = M._conv_mat(rng, nf, nfilter, winlen)
    bias = rng.uniform(-0.1, 0.1, nfilter).astype(np.float32)
    layer = M.ConvLayer(cm, M.Mat.vector(bias), stride, nf, winlen)
    x = rng.standard_normal((T, nf)).astype(np.float32)
    got = ffo.take(ffo.lib().fo_convolution(ffo.HostMat.from_dense(x).ptr, ffo.HostMat.from_model_mat(cm).ptr,
                                            ffo.HostMat.from_model_mat(layer.b).ptr, stride))
    nv = naive_conv(x, layer.taps(), bias, stride)
    assert np.abs(got[:798] - nv[:798]).max() <= 2e-5
    assert np.abs(got[798] - nv[799]).max() <= 2e-5
    assert np.array_equal(got[799], bias)


# ---- recurrent layers / CRF against float64 numpy ---------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


@pytest.mark.parametrize("backward", [0, 1])
def test_lstm_against_numpy(backward):
    rng = np.random.default_rng(3)
    H, n = 32, 50
    xa = rng.standard_normal((n, 4 * H)).astype(np.float32)
    sW = (rng.standard_normal((4 * H, H)) / np.sqrt(H)).astype(np.float32)
    got = ffo.take(ffo.lib().fo_lstm(ffo.HostMat.from_dense(xa).ptr, ffo.HostMat.from_dense(sW).ptr, backward))
    h, c = np.zeros(H), np.zeros(H)
    want = np.zeros((n, H))
    order = range(n - 1, -1, -1) if backward else range(n)
    for t in order:
        g = xa[t].astype(np.float64) + sW.astype(np.float64) @ h
        i, f, gg, o = g[:H], g[H:2 * H], g[2 * H:3 * H], g[3 * H:]
        c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(gg)
        h = _sigmoid(o) * np.tanh(c)
        want[t] = h
    assert np.abs(got - want).max() <= 5e-6


@pytest.mark.parametrize("backward", [0, 1])
def test_grumod_against_numpy(backward):
    rng = np.random.default_rng(4)
    H, n = 32, 50
    x = rng.standard_normal((n, 3 * H)).astype(np.float32)
    sW = (rng.standard_normal((3 * H, H)) / np.sqrt(H)).astype(np.float32)
    got = ffo.take(ffo.lib().fo_grumod(ffo.HostMat.from_dense(x).ptr, ffo.HostMat.from_dense(sW).ptr, backward))
    h = np.zeros(H)
    want = np.zeros((n, H))
    order = range(n - 1, -1, -1) if backward else range(n)
    for t in order:
        s = sW.astype(np.float64) @ h
        z = _sigmoid(x[t, :H] + s[:H])
        r = _sigmoid(x[t, H:2 * H] + s[H:2 * H])
        hbar = np.tanh(r * s[2 * H:] + x[t, 2 * H:])
        h = z * h + (1 - z) * hbar
        want[t] = h
    assert np.abs(got - want).max() <= 5e-6


def _logsumexp(a):
    m = np.max(a)
    return m + np.log(np.sum(np.exp(a - m)))


@pytest.mark.parametrize("nbase", [4, 5])
def test_partition_and_posterior_against_bruteforce(nbase):
    """Partition function = log-sum over all state paths; posteriors sum to one per block."""
    rng = np.random.default_rng(9)
    ns, P, nblk = 2 * nbase, 2 * nbase * (nbase + 1), 6
    S = rng.standard_normal((nblk, P)).astype(np.float32)
    got = ffo.lib().fo_partition_function(ffo.HostMat.from_dense(S).ptr)
    # dense transition matrix per block: score[to][from] or -inf if not allowed
    def dense(row):
        Tm = np.full((ns, ns), -np.inf)
        for to in range(nbase):
            Tm[to, :] = row[to * ns: to * ns + ns]
        for b in range(nbase, ns):
            Tm[b, b] = row[nbase * ns + b]
            Tm[b, b - nbase] = row[nbase * ns + b - nbase]
        return Tm
    v = np.zeros(ns)
    for blk in range(nblk):
        Tm = dense(S[blk].astype(np.float64))
        v = np.array([_logsumexp(Tm[to] + v) for to in range(ns)])
    assert abs(got - _logsumexp(v)) <= 1e-9
    post = ffo.take(ffo.lib().fo_transpost(ffo.HostMat.from_dense(S).ptr, 0))
    assert np.abs(post.sum(axis=1) - 1.0).max() <= 1e-5


def test_viterbi_against_bruteforce():
    rng = np.random.default_rng(10)
    nbase, nblk = 4, 5
    ns, P = 8, 40
    S = rng.standard_normal((nblk, P)).astype(np.float32)
    path = np.zeros(nblk + 1, dtype=np.int32)
    qpath = np.zeros(nblk + 1, dtype=np.float32)
    score = ffo.lib().fo_decode_viterbi(ffo.HostMat.from_dense(S).ptr, 0, path.ctypes.data_as(C.POINTER(C.c_int)), _f(qpath))
    # exhaustive search
    import itertools
    best = -np.inf
    for p in itertools.product(range(ns), repeat=nblk + 1):
        tot = 0.0
        ok = True
        for blk in range(nblk):
            fr, to = p[blk], p[blk + 1]
            if to < nbase:
                tot += S[blk, to * ns + fr]
            elif fr == to or fr == to - nbase:
                tot += S[blk, nbase * ns + fr]
            else:
                ok = False
                break
        if ok and tot > best:
            best = tot
    assert abs(score - best) <= 1e-4
    assert np.isnan(qpath[0])
    assert abs(float(np.sum(qpath[1:])) - score) <= 1e-4


# ---- regression vectors ------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["lstm5_h64", "grumod5_h64", "lstm5_h96_t1237", "lstm5_h256_t1500", "lstm5_h384_t1500", "lstm5_h512_t1000", "grumod5_h256_t1000"])
def test_oracle_regression_vectors(tag, golden_dir):
    g = np.load(os.path.join(golden_dir, "oracle_net_%s.npz" % tag))
    mdl = M.synthetic_model(int(g["kind"]), int(g["hidden"]), seed=int(g["seed"]))
    om = ffo.OracleModel(mdl)
    for i in (0, 1):
        res = om.basecall(g["signal%d" % i])
        assert np.array_equal(res["trans"], g["trans%d" % i])
        assert np.array_equal(res["path"], g["path%d" % i])
        assert res["basecall"].encode() == g["basecall%d" % i].tobytes()
        assert res["quality"].encode() == g["quality%d" % i].tobytes()
        assert np.array_equal(res["trace"], g["trace%d" % i].astype(np.int32))
        # invariants: trace columns are probabilities x 255
        assert res["trace"].min() >= 0 and res["trace"].max() <= 255


def test_dot_modes_stay_within_rounding_of_the_oracle():
    """ff_oracle.c's other summation modes (1: double accumulator, the yardstick of tests/test_fuzz_tail_gpu.py; 2: the
    vectorised kernels of cpu_ref.c that bench.py's cpu_baseline times) evaluate the same network: identical calls and
    transition scores within a few 1e-5 of the reference-order oracle on both model families."""
    for kind, hidden, T in ((M.NET_LSTM5, 96, 1505), (M.NET_GRUMOD5, 64, 1200), (M.NET_LSTM5, 36, 600)):
        mdl = M.synthetic_model(kind, hidden, seed=3)
        om = ffo.OracleModel(mdl)
        sig = np.random.default_rng(hidden).standard_normal(T).astype(np.float32)
        ref = om.basecall(sig)
        for mode in (1, 2):
            with ffo.dot_mode(mode):
                alt = om.basecall(sig)
            assert np.abs(alt["trans"] - ref["trans"]).max() <= 5e-5, (kind, hidden, mode)
            assert alt["basecall"] == ref["basecall"]
        assert np.array_equal(om.basecall(sig)["trans"], ref["trans"])      # mode restored


def test_openblas_dot_mode_matches_the_oracle():
    """Dot mode 3 -- the GEMV / GEMM calls of the reference (layers.c:1009, :250, flappie_matrix.c:384) made in a real OpenBLAS found on
    this host through dlopen (bench.py's cpu_baseline, kind "port+openblas") -- is the same network: transition scores within a few
    1e-5 of the reference-order oracle, identical calls.  Skipped where no LP64 OpenBLAS exists."""
    import bench
    mode, blas = bench._oracle_with_blas(3)
    ffo.lib().fo_set_dot_mode(0)
    if blas is None:
        pytest.skip("no LP64 OpenBLAS on this host")
    assert mode == 3 and "OpenBLAS" in blas[1]
    for kind, hidden, T in ((M.NET_LSTM5, 96, 1505), (M.NET_GRUMOD5, 64, 1200)):
        mdl = M.synthetic_model(kind, hidden, seed=3)
        om = ffo.OracleModel(mdl)
        sig = np.random.default_rng(hidden).standard_normal(T).astype(np.float32)
        ref = om.basecall(sig)
        with ffo.dot_mode(3):
            alt = om.basecall(sig)
        assert np.abs(alt["trans"] - ref["trans"]).max() <= 5e-5, (kind, hidden)
        assert alt["basecall"] == ref["basecall"]
