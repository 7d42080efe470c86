"""The reference's per-layer operator interface (include/layers.h, flappie_matrix.h compute entries) over the
HIP engine, checked against the oracle layer by layer -- the shape of the reference's own unit tests
(src/test/test_flappie_convolution.c, test_flappie_elu.c, test_flappie_matrix.c).  CPU tests cover the
host-only helpers and the oracle's new element-wise functions; GPU tests call the C host library, which
runs the batch-of-one form of the production kernels."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ffo
from test_host_layer import CMat, HOSTLIB, _f

PM = C.POINTER(CMat)


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(HOSTLIB):
        pytest.fail("libflappie_host.so not built: run __graft_entry__.build()")
    lib = C.CDLL(HOSTLIB)
    lib.make_flappie_matrix.restype = PM
    lib.make_flappie_matrix.argtypes = [C.c_size_t, C.c_size_t]
    lib.mat_from_array.restype = PM
    lib.mat_from_array.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t]
    lib.free_flappie_matrix.restype = PM
    lib.free_flappie_matrix.argtypes = [PM]
    for name in ("swish", "tanh", "exp", "log", "elu"):
        getattr(lib, name + "_activation_inplace").argtypes = [PM]
        getattr(lib, name + "_activation_inplace").restype = None
    lib.robustlog_activation_inplace.argtypes = [PM, C.c_float]
    lib.shift_scale_matrix_inplace.argtypes = [PM, C.c_float, C.c_float]
    lib.row_normalise_inplace.argtypes = [PM]
    lib.log_row_normalise_inplace.argtypes = [PM]
    lib.residual_inplace.argtypes = [PM, PM]
    lib.residual.restype = PM
    lib.residual.argtypes = [PM, PM, PM]
    lib.embedding.restype = PM
    lib.embedding.argtypes = [C.POINTER(C.c_int), C.c_size_t, PM, PM]
    lib.window.restype = PM
    lib.window.argtypes = [PM, C.c_size_t, C.c_size_t]
    lib.convolution.restype = PM
    lib.convolution.argtypes = [PM, PM, PM, C.c_size_t, PM]
    for name in ("affine_map", "feedforward_linear", "feedforward_tanh", "feedforward_exp", "softmax"):
        getattr(lib, name).restype = PM
        getattr(lib, name).argtypes = [PM, PM, PM, PM]
    lib.softmax_with_temperature.restype = PM
    lib.softmax_with_temperature.argtypes = [PM, PM, PM, C.c_float, C.c_float, PM]
    for name in ("affine_map2", "feedforward2_tanh"):
        getattr(lib, name).restype = PM
        getattr(lib, name).argtypes = [PM] * 6
    for name in ("lstm_forward", "lstm_backward", "grumod_forward", "grumod_backward"):
        getattr(lib, name).restype = PM
        getattr(lib, name).argtypes = [PM, PM, PM]
    lib.lstm_step.argtypes = [PM] * 6
    lib.lstm_step.restype = None
    lib.grumod_step.argtypes = [PM] * 5
    lib.grumod_step.restype = None
    for name in ("gru_forward", "gru_backward", "gru_relu_forward", "gru_relu_backward"):
        getattr(lib, name).restype = PM
        getattr(lib, name).argtypes = [PM, PM, PM, PM]
    for name in ("gru_step", "gru_relu_step"):
        getattr(lib, name).argtypes = [PM] * 6
        getattr(lib, name).restype = None
    lib.crf_manystay_partition_function.restype = C.c_double
    lib.crf_manystay_partition_function.argtypes = [PM]
    for name in ("globalnorm_flipflop", "globalnorm_manystay"):
        getattr(lib, name).restype = PM
        getattr(lib, name).argtypes = [PM, PM, PM, C.c_float, PM]
    lib.nbase_from_flipflop_nparam.restype = C.c_size_t
    lib.nbase_from_flipflop_nparam.argtypes = [C.c_size_t]
    lib.min_flappie_matrix.restype = C.c_float
    lib.min_flappie_matrix.argtypes = [PM]
    lib.max_flappie_matrix.restype = C.c_float
    lib.max_flappie_matrix.argtypes = [PM]
    lib.validate_flappie_matrix.restype = C.c_bool
    lib.validate_flappie_matrix.argtypes = [PM, C.c_float, C.c_float, C.c_float, C.c_bool, C.c_char_p, C.c_int]
    lib.clip_matrix_inplace.argtypes = [PM, C.c_float]
    lib.filter_matrix_inplace.argtypes = [PM, C.c_float, C.c_float]
    lib.difference_matrix_inplace.argtypes = [PM, C.c_float]
    return lib


def mk(L, a):
    """dense [nc, nr] -> flappie_matrix"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    nc, nr = a.shape
    return L.mat_from_array(_f(a), nr, nc)


def dense(pm):
    m = pm.contents
    return np.ctypeslib.as_array(m.f, shape=(m.nc, m.stride))[:, : m.nr].copy()


def image(pm):
    m = pm.contents
    return np.ctypeslib.as_array(m.f, shape=(m.nc, m.stride)).copy()


def omat(a):
    return ffo.HostMat.from_dense(a)


# ------------------------------------------------------------------------------------ CPU
def test_oracle_log_cephes_against_libm():
    lib = ffo.lib()
    rng = np.random.default_rng(4)
    xs = np.concatenate([np.exp(rng.uniform(-80, 80, 4000)), [1.0, 0.5, 2.0, 1.17549435e-38, 3.4e38]]).astype(np.float32)
    got = np.array([lib.fo_logf_cephes(float(x)) for x in xs], dtype=np.float32)
    want = np.log(xs.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got - want) / np.maximum(ulp, 1e-45)) <= 2.5          # cephes logf: < 2 ulp
    assert lib.fo_logf_cephes(1.0) == 0.0
    assert np.isnan(lib.fo_logf_cephes(0.0)) and np.isnan(lib.fo_logf_cephes(-1.0))     # invalid mask, sse_mathfun.h:131,206
    assert lib.fo_logf_cephes(1e-42) == lib.fo_logf_cephes(1.17549435e-38)             # denormals clamped, :133


def test_oracle_elementwise_and_affine2():
    lib = ffo.lib()
    rng = np.random.default_rng(5)
    a = rng.standard_normal((7, 6)).astype(np.float32)
    m = omat(a)
    lib.fo_elu_inplace(m.ptr)
    want = np.where(a >= 0, a, np.expm1(a.astype(np.float64)))
    np.testing.assert_allclose(m.data[:, :6], want, rtol=0, atol=2e-7)
    assert np.all(m.data[:, 6:] == 0)                                           # elu(0) = 0 on the pad lanes
    p = rng.uniform(0, 1, (7, 6)).astype(np.float32)
    m = omat(p)
    lib.fo_robustlog_inplace(m.ptr, 0.01)
    np.testing.assert_allclose(m.data[:, :6], np.log(0.01 + 0.99 * p.astype(np.float64)), rtol=0, atol=3e-7)
    xf, xb = rng.standard_normal((9, 10)).astype(np.float32), rng.standard_normal((9, 6)).astype(np.float32)
    wf, wb = rng.standard_normal((5, 10)).astype(np.float32), rng.standard_normal((5, 6)).astype(np.float32)
    b = rng.standard_normal((1, 5)).astype(np.float32)
    got = ffo.take(lib.fo_affine_map2(omat(xf).ptr, omat(xb).ptr, omat(wf).ptr, omat(wb).ptr, omat(b).ptr))
    np.testing.assert_allclose(got, xf @ wf.T + xb @ wb.T + b, rtol=0, atol=1e-5)


def test_host_only_matrix_helpers(L):
    rng = np.random.default_rng(6)
    a = rng.standard_normal((5, 7)).astype(np.float32)
    m = mk(L, a)
    assert L.max_flappie_matrix(m) == a.max() and L.min_flappie_matrix(m) == a.min()
    assert L.validate_flappie_matrix(m, np.nan, np.nan, 0.0, True, b"t", 1)
    assert not L.validate_flappie_matrix(m, 0.0, np.nan, 0.0, True, b"t", 1)            # lower bound violated
    m.contents.f[7] = 1.0                                                               # pad lane of column 0
    assert not L.validate_flappie_matrix(m, np.nan, np.nan, 0.0, True, b"t", 1)         # masking rule
    m.contents.f[7] = 0.0
    L.clip_matrix_inplace(m, 0.5)
    np.testing.assert_array_equal(dense(m), np.clip(a, -0.5, 0.5))
    L.filter_matrix_inplace(m, 9.0, 0.4)
    want = np.where(np.abs(np.clip(a, -0.5, 0.5)) > 0.4, np.float32(9.0), np.clip(a, -0.5, 0.5))
    np.testing.assert_array_equal(dense(m), want)
    L.difference_matrix_inplace(m, -1.0)
    d = np.vstack([want[1:] - want[:-1], np.full((1, 7), -1.0, np.float32)])
    np.testing.assert_array_equal(dense(m), d)
    L.free_flappie_matrix(m)
    # embedding / window: data movement (layers.c:127-176)
    E = rng.standard_normal((4, 6)).astype(np.float32)
    idx = np.array([3, 0, 0, 2, 1], dtype=np.int32)
    e = L.embedding(idx.ctypes.data_as(C.POINTER(C.c_int)), 5, mk(L, E), None)
    np.testing.assert_array_equal(dense(e), E[idx])
    assert not L.embedding(None, 5, mk(L, E), None)
    x = rng.standard_normal((9, 3)).astype(np.float32)
    w = L.window(mk(L, x), 4, 2)
    got = dense(w)
    assert got.shape == (5, 12)
    for col in range(5):
        for k, w1 in enumerate(range(2 * col - 1, 2 * col + 3)):
            want = x[w1] if 0 <= w1 < 9 else np.zeros(3, np.float32)
            np.testing.assert_array_equal(got[col, 3 * k:3 * k + 3], want)
    assert L.nbase_from_flipflop_nparam(40) == 4 and L.nbase_from_flipflop_nparam(60) == 5


def test_null_inputs_propagate(L):
    """RETURN_NULL_IF semantics (flappie_stdlib.h:44): a NULL input is a NULL output, never a crash."""
    assert not L.convolution(None, None, None, 1, None)
    assert not L.affine_map(None, None, None, None)
    assert not L.lstm_forward(None, None, None)
    assert not L.grumod_backward(None, None, None)
    assert not L.globalnorm_flipflop(None, None, None, 1.0, None)
    assert np.isnan(L.crf_manystay_partition_function(None))
    L.tanh_activation_inplace(None)
    L.row_normalise_inplace(None)
    L.residual_inplace(None, None)


# ------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_activations_bit_exact_including_pad_lanes(L):
    lib = ffo.lib()
    rng = np.random.default_rng(10)
    a = np.concatenate([rng.standard_normal((40, 6)) * 4, [[0, -0.0, 88.5, -88.5, 100, -100]]]).astype(np.float32)
    for name, ofn in (("swish", lib.fo_swish_inplace), ("tanh", lib.fo_tanh_inplace), ("exp", lib.fo_exp_inplace),
                      ("elu", lib.fo_elu_inplace)):
        m, o = mk(L, a), omat(a)
        getattr(L, name + "_activation_inplace")(m)
        ofn(o.ptr)
        np.testing.assert_array_equal(image(m), o.data, err_msg=name)
        L.free_flappie_matrix(m)
    p = np.abs(a) + np.float32(1e-3)
    m, o = mk(L, p), omat(p)
    L.log_activation_inplace(m)
    lib.fo_log_inplace(o.ptr)
    np.testing.assert_array_equal(image(m), o.data)               # pad lanes: log(0) = NaN in both
    L.free_flappie_matrix(m)
    q = rng.uniform(0, 1, (12, 5)).astype(np.float32)
    m, o = mk(L, q), omat(q)
    L.robustlog_activation_inplace(m, 0.02)
    lib.fo_robustlog_inplace(o.ptr, 0.02)
    np.testing.assert_array_equal(image(m), o.data)
    L.free_flappie_matrix(m)
    m = mk(L, a)
    L.shift_scale_matrix_inplace(m, 0.25, 3.0)
    want = np.zeros_like(image(m))
    want[:, :6] = (a - np.float32(0.25)) / np.float32(3.0)        # rows < nr only (flappie_matrix.c:625-633)
    np.testing.assert_array_equal(image(m), want)
    L.free_flappie_matrix(m)


CONV_CASES = [(1, 4, 5, 1, 100), (4, 16, 5, 1, 101), (16, 96, 19, 5, 400), (16, 96, 19, 5, 403), (16, 64, 19, 5, 399),
              (1, 48, 19, 2, 105), (16, 40, 19, 3, 100), (16, 8, 7, 5, 57), (3, 6, 11, 2, 64), (16, 33, 20, 5, 100),
              (1, 4, 5, 1, 5), (16, 96, 19, 5, 19), (2, 2, 1, 1, 7)]


@pytest.mark.gpu
@pytest.mark.parametrize("nf,nfilter,winlen,stride,T", CONV_CASES)
def test_convolution_matches_oracle(L, nf, nfilter, winlen, stride, T):
    """layers.c:189-276 including the strided right-edge behaviour; thin layers take the VALU kernel, wide ones
    the MFMA implicit-GEMM kernel -- the two kernels of the production path."""
    rng = np.random.default_rng(nf * 1000 + winlen * 10 + stride + T)
    nfp = 4 * ((nf + 3) // 4)
    x = rng.standard_normal((T, nf)).astype(np.float32)
    w = np.zeros((nfilter, winlen * nfp), dtype=np.float32)
    for t in range(winlen):
        w[:, t * nfp:t * nfp + nf] = rng.standard_normal((nfilter, nf)) / np.sqrt(nf * winlen)
    nrw = nfp * winlen - nfp + nf                                   # taiyaki_flipflop5_guppy.py:92-94
    w = w[:, :nrw]
    b = rng.standard_normal((1, nfilter)).astype(np.float32)
    want = ffo.take(ffo.lib().fo_convolution(omat(x).ptr, omat(w).ptr, omat(b).ptr, stride))
    X, W, Bm = mk(L, x), mk(L, w), mk(L, b)
    c = L.convolution(X, W, Bm, stride, None)
    assert c, "convolution returned NULL"
    got = dense(c)
    assert got.shape == want.shape == ((T + stride - 1) // stride, nfilter)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    assert np.all(image(c)[:, nfilter:] == 0)
    # output reuse: same shape -> same object; different shape -> reallocated (flappie_matrix.c:54-61)
    c2 = L.convolution(X, W, Bm, stride, c)
    assert C.addressof(c2.contents) == C.addressof(c.contents)
    np.testing.assert_array_equal(dense(c2), got)
    for m in (X, W, Bm, c2):
        L.free_flappie_matrix(m)


@pytest.mark.gpu
def test_identity_convolution(L):
    """test_flappie_convolution.c:395-416: a one-tap identity filter returns its input."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((50, 4)).astype(np.float32)
    c = L.convolution(mk(L, x), mk(L, np.eye(4, dtype=np.float32)), mk(L, np.zeros((1, 4), np.float32)), 1, None)
    np.testing.assert_array_equal(dense(c), x)


@pytest.mark.gpu
@pytest.mark.parametrize("K,M,T", [(384, 1536, 100), (96, 40, 37), (36, 60, 16), (5, 3, 1), (256, 768, 250)])
def test_feedforward_and_softmax(L, K, M, T):
    lib = ffo.lib()
    rng = np.random.default_rng(K + M + T)
    x = rng.standard_normal((T, K)).astype(np.float32)
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal((1, M)).astype(np.float32)
    X, W, Bm = mk(L, x), mk(L, w), mk(L, b)
    want = ffo.take(lib.fo_affine_map(omat(x).ptr, omat(w).ptr, omat(b).ptr))
    for fn in ("affine_map", "feedforward_linear"):
        c = getattr(L, fn)(X, W, Bm, None)
        np.testing.assert_allclose(dense(c), want, rtol=0, atol=3e-5)
        L.free_flappie_matrix(c)
    c = L.feedforward_tanh(X, W, Bm, None)
    o = omat(want); lib.fo_tanh_inplace(o.ptr)
    np.testing.assert_allclose(dense(c), o.data[:, :M], rtol=0, atol=3e-5)
    c = L.feedforward_exp(X, W, Bm, c)
    o = omat(want); lib.fo_exp_inplace(o.ptr)
    np.testing.assert_allclose(dense(c), o.data[:, :M], rtol=3e-5, atol=1e-6)
    assert np.all(image(c)[:, M:] == 1.0)                          # exp over the pad lanes (layers.c:56-66)
    c = L.softmax(X, W, Bm, c)
    lib.fo_row_normalise_inplace(o.ptr)
    np.testing.assert_allclose(dense(c), o.data[:, :M], rtol=5e-5, atol=1e-7)
    np.testing.assert_allclose(dense(c).sum(axis=1), 1.0, rtol=0, atol=1e-5)
    # softmax_with_temperature scales X in place (layers.c:376-395)
    X2 = mk(L, x)
    c2 = L.softmax_with_temperature(X2, W, Bm, 2.0, 0.5, None)
    np.testing.assert_array_equal(dense(X2), x / np.float32(4.0))
    z = (x.astype(np.float64) / 4.0) @ w.T.astype(np.float64) + b
    z = z / 0.5
    sm = np.exp(z - z.max(axis=1, keepdims=True))
    sm /= sm.sum(axis=1, keepdims=True)
    np.testing.assert_allclose(dense(c2), sm, rtol=2e-4, atol=1e-6)
    # row_normalise_inplace: the reference's own arithmetic order -> bit-exact against the oracle
    r = np.abs(rng.standard_normal((T, M))).astype(np.float32) + np.float32(0.1)
    R, o = mk(L, r), omat(r)
    L.row_normalise_inplace(R)
    lib.fo_row_normalise_inplace(o.ptr)
    np.testing.assert_array_equal(dense(R), o.data[:, :M])
    R, o = mk(L, want), omat(want)
    L.log_row_normalise_inplace(R)
    lib.fo_log_row_normalise_inplace(o.ptr)
    np.testing.assert_allclose(dense(R), o.data[:, :M], rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_feedforward2_and_residual(L):
    lib = ffo.lib()
    rng = np.random.default_rng(12)
    xf, xb = rng.standard_normal((33, 96)).astype(np.float32), rng.standard_normal((33, 40)).astype(np.float32)
    wf = (rng.standard_normal((24, 96)) / 10).astype(np.float32)
    wb = (rng.standard_normal((24, 40)) / 6).astype(np.float32)
    b = rng.standard_normal((1, 24)).astype(np.float32)
    want = ffo.take(lib.fo_affine_map2(omat(xf).ptr, omat(xb).ptr, omat(wf).ptr, omat(wb).ptr, omat(b).ptr))
    c = L.affine_map2(mk(L, xf), mk(L, xb), mk(L, wf), mk(L, wb), mk(L, b), None)
    np.testing.assert_allclose(dense(c), want, rtol=0, atol=3e-5)
    c = L.feedforward2_tanh(mk(L, xf), mk(L, xb), mk(L, wf), mk(L, wb), mk(L, b), c)
    o = omat(want); lib.fo_tanh_inplace(o.ptr)
    np.testing.assert_allclose(dense(c), o.data[:, :24], rtol=0, atol=3e-5)
    a, f = rng.standard_normal((20, 10)).astype(np.float32), rng.standard_normal((20, 10)).astype(np.float32)
    r = L.residual(mk(L, a), mk(L, f), None)
    np.testing.assert_array_equal(dense(r), a + f)
    F = mk(L, f)
    L.residual_inplace(mk(L, a), F)
    np.testing.assert_array_equal(dense(F), a + f)


RNN_SHAPES = [(32, 50), (96, 77), (36, 23), (256, 40), (384, 30)]


@pytest.mark.gpu
@pytest.mark.parametrize("H,T", RNN_SHAPES)
@pytest.mark.parametrize("kind", ["lstm", "grumod"])
def test_recurrent_layers_match_oracle(L, kind, H, T):
    """lstm_forward/backward (layers.c:877-976), grumod_forward/backward (layers.c:571-660)."""
    lib = ffo.lib()
    G = 4 if kind == "lstm" else 3
    rng = np.random.default_rng(H * 7 + T + G)
    xa = rng.standard_normal((T, G * H)).astype(np.float32)
    sw = (rng.standard_normal((G * H, H)) / np.sqrt(H)).astype(np.float32)
    X, S = mk(L, xa), mk(L, sw)
    ofn = lib.fo_lstm if kind == "lstm" else lib.fo_grumod
    for direction, backward in (("forward", 0), ("backward", 1)):
        want = ffo.take(ofn(omat(xa).ptr, omat(sw).ptr, backward))
        out = getattr(L, "%s_%s" % (kind, direction))(X, S, None)
        assert out, "%s_%s returned NULL" % (kind, direction)
        got = dense(out)
        assert got.shape == (T, H)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
        L.free_flappie_matrix(out)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lstm", "grumod"])
def test_recurrent_steps_chain_to_the_layer(L, kind):
    """lstm_step (layers.c:979-1026) / grumod_step (layers.c:664-715) applied T times from a zero state
    reproduce lstm_forward / grumod_forward."""
    H, T = 48, 9
    G = 4 if kind == "lstm" else 3
    rng = np.random.default_rng(31 + G)
    xa = rng.standard_normal((T, G * H)).astype(np.float32)
    sw = (rng.standard_normal((G * H, H)) / np.sqrt(H)).astype(np.float32)
    S = mk(L, sw)
    layer = dense(getattr(L, kind + "_forward")(mk(L, xa), S, None))
    want = ffo.take((ffo.lib().fo_lstm if kind == "lstm" else ffo.lib().fo_grumod)(omat(xa).ptr, omat(sw).ptr, 0))
    h = mk(L, np.zeros((1, H), np.float32))
    state = mk(L, np.zeros((1, H), np.float32))
    xF = L.make_flappie_matrix(G * H, 1)
    for t in range(T):
        x = mk(L, xa[t:t + 1])
        out = L.make_flappie_matrix(H, 1)
        if kind == "lstm":
            L.lstm_step(x, h, S, xF, state, out)
        else:
            L.grumod_step(x, h, S, xF, out)
        np.testing.assert_allclose(dense(out)[0], want[t], rtol=0, atol=2e-5)
        np.testing.assert_allclose(dense(out)[0], layer[t], rtol=0, atol=2e-5)
        L.free_flappie_matrix(h)
        L.free_flappie_matrix(x)
        h = out


def _gru_numpy(xa, sw, sw2, backward, relu):
    """the sloika GRU in float64 numpy, written from the equations (an independent check of the oracle's restatement)"""
    T, H = xa.shape[0], sw2.shape[0]
    x, w, w2 = xa.astype(np.float64), sw.astype(np.float64), sw2.astype(np.float64)
    out, h = np.zeros((T, H)), np.zeros(H)
    for i in range(T):
        t = T - 1 - i if backward else i
        zr = 1.0 / (1.0 + np.exp(-(x[t, :2 * H] + w @ h)))
        z, r = zr[:H], zr[H:]
        pre = x[t, 2 * H:] + w2 @ (r * h)
        hbar = np.maximum(pre, 0.0) if relu else np.tanh(pre)
        h = z * h + (1.0 - z) * hbar
        out[t] = h
    return out


@pytest.mark.parametrize("relu", [0, 1])
def test_oracle_sloika_gru_against_numpy(relu):
    """fo_gru (layers.c:412-568, 718-874) against the equations in float64"""
    H, T = 24, 40
    rng = np.random.default_rng(5 + relu)
    xa = rng.standard_normal((T, 3 * H)).astype(np.float32)
    sw = (rng.standard_normal((2 * H, H)) / np.sqrt(H)).astype(np.float32)
    sw2 = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32)
    for backward in (0, 1):
        got = ffo.take(ffo.lib().fo_gru(omat(xa).ptr, omat(sw).ptr, omat(sw2).ptr, backward, relu, None))
        np.testing.assert_allclose(got, _gru_numpy(xa, sw, sw2, backward, relu), rtol=0, atol=5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("H,T", [(32, 50), (96, 77), (36, 23), (256, 40)])
@pytest.mark.parametrize("relu", [0, 1])
def test_sloika_gru_layers_match_oracle(L, relu, H, T):
    """gru_forward/backward (layers.c:412-510), gru_relu_forward/backward (layers.c:718-816)"""
    rng = np.random.default_rng(H * 3 + T + relu)
    xa = rng.standard_normal((T, 3 * H)).astype(np.float32)
    sw = (rng.standard_normal((2 * H, H)) / np.sqrt(H)).astype(np.float32)
    sw2 = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32)
    X, S, S2 = mk(L, xa), mk(L, sw), mk(L, sw2)
    stem = "gru_relu" if relu else "gru"
    for direction, backward in (("forward", 0), ("backward", 1)):
        want = ffo.take(ffo.lib().fo_gru(omat(xa).ptr, omat(sw).ptr, omat(sw2).ptr, backward, relu, None))
        out = getattr(L, "%s_%s" % (stem, direction))(X, S, S2, None)
        assert out, "%s_%s returned NULL" % (stem, direction)
        got = dense(out)
        assert got.shape == (T, H)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
        L.free_flappie_matrix(out)
    assert not L.gru_forward(None, S, S2, None)
    assert not L.gru_forward(X, S, mk(L, sw), None)          # sW2 of the wrong shape: refused


@pytest.mark.gpu
@pytest.mark.parametrize("relu", [0, 1])
def test_sloika_gru_steps_chain_to_the_layer(L, relu):
    """gru_step (layers.c:513-568) / gru_relu_step (layers.c:819-874) applied T times from a zero state"""
    H, T = 48, 9
    rng = np.random.default_rng(77 + relu)
    xa = rng.standard_normal((T, 3 * H)).astype(np.float32)
    sw = (rng.standard_normal((2 * H, H)) / np.sqrt(H)).astype(np.float32)
    sw2 = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float32)
    S, S2 = mk(L, sw), mk(L, sw2)
    want = ffo.take(ffo.lib().fo_gru(omat(xa).ptr, omat(sw).ptr, omat(sw2).ptr, 0, relu, None))
    h = mk(L, np.zeros((1, H), np.float32))
    xF = L.make_flappie_matrix(3 * H, 1)
    for t in range(T):
        x = mk(L, xa[t:t + 1])
        out = L.make_flappie_matrix(H, 1)
        (L.gru_relu_step if relu else L.gru_step)(x, h, S, S2, xF, out)
        np.testing.assert_allclose(dense(out)[0], want[t], rtol=0, atol=2e-5)
        L.free_flappie_matrix(h)
        L.free_flappie_matrix(x)
        h = out


@pytest.mark.gpu
@pytest.mark.parametrize("nbase,H,T", [(4, 96, 200), (5, 64, 333), (4, 384, 50), (2, 36, 17), (4, 32, 1)])
def test_globalnorm_and_partition_function(L, nbase, H, T):
    lib = ffo.lib()
    P = 2 * nbase * (nbase + 1)
    rng = np.random.default_rng(nbase + H + T)
    x = np.tanh(rng.standard_normal((T, H))).astype(np.float32)
    w = (rng.standard_normal((P, H)) / np.sqrt(H) * 3).astype(np.float32)
    b = rng.standard_normal((1, P)).astype(np.float32)
    for temperature in (1.0, 0.7):
        want = ffo.take(lib.fo_globalnorm_flipflop(omat(x).ptr, omat(w).ptr, omat(b).ptr, temperature))
        c = L.globalnorm_flipflop(mk(L, x), mk(L, w), mk(L, b), temperature, None)
        assert c
        np.testing.assert_allclose(dense(c), want, rtol=0, atol=5e-5)
        # a globally normalised score matrix has partition function 0 (up to the float rounding of logZ/T)
        z = L.crf_manystay_partition_function(c)
        assert abs(z) <= 1e-5 * T + 1e-4
    s = (rng.standard_normal((T, P)) * 2).astype(np.float32)
    want = lib.fo_partition_function(omat(s).ptr)
    got = L.crf_manystay_partition_function(mk(L, s))
    assert abs(got - want) <= 1e-9 * max(1.0, abs(want))
    assert L.nbase_from_flipflop_nparam(P) == nbase
    # the pipeline's scaled linear-space recursion gives the same fp64 number (ffhip_decode.hip: the forward chain of k_crf_fb; bounds beyond kFbRange / 2: k_crf_chain)
    from flappie_amd import binding as B

    class FMat(C.Structure):
        _fields_ = [("data", C.POINTER(C.c_float)), ("nr", C.c_size_t), ("nc", C.c_size_t), ("stride", C.c_size_t),
                    ("dev", C.c_void_p), ("dev_state", C.c_void_p)]          # (a plain host array: no device image)
    eng = B.Engine(0)
    fn = B.lib().ffhip_op_partition_function_scaled
    fn.argtypes = [C.c_void_p, FMat, C.c_float, C.POINTER(C.c_double)]
    for scale in (1.0, 5.0, 20.0):
        s2 = np.clip(s * scale / 2, -scale * 2.5, scale * 2.5).astype(np.float32)
        o = omat(s2)
        want = lib.fo_partition_function(o.ptr)
        z = C.c_double(0)
        assert fn(eng.h, FMat(o.c.f, P, T, o.c.stride), float(np.abs(s2).max()), C.byref(z)) == 0
        assert abs(z.value - want) <= 1e-11 * max(1.0, abs(want)), (scale, z.value, want)
    assert fn(eng.h, FMat(o.c.f, P, T, o.c.stride), 1e4, C.byref(z)) != 0          # bound too wide: refused, not wrong
    eng.close()
