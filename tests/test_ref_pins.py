"""The oracle held to the REFERENCE's own object code, where that builds in this image (oracle/_ref, made by
oracle/Makefile from /root/reference/src in the build container; the .so files travel, the sources do not):

  * libflappie_inlref.so -- oracle/ref_inline.c: loops over the reference's header-inline arithmetic
    (util.h:276-346, sse_mathfun.h:123-301).  The oracle's scalar restatements must agree BIT FOR BIT on > 10^7
    inputs per function, clamps, denormals, infinities and NaNs included.  Pins row A4 and the scalar pieces of
    A9 / A10 / A12.
  * libflappie_decref.so -- the reference's decode.c + util.c compiled unchanged, with oracle/ref_decode_glue.c for
    the allocation / normalisation symbols of the unbuildable flappie_matrix.c / layers.c.  The oracle's Viterbi,
    forward/backward posterior, trace, change positions and the run-length decoders must agree bit for bit on random
    and on tie-heavy inputs.  Pins rows A10-A13 (and N4's decoders) through a partial reference build with declared glue.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ffo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = C.POINTER


def _load(name):
    path = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/%s not built (reference sources absent on this machine)" % name)
    return C.CDLL(path)


def _f(a):
    return a.ctypes.data_as(P(C.c_float))


def _sweep(lo=None, hi=None):
    """> 1.2e7 float32 inputs: every exponent through random bit patterns, a dense linear sweep, and the specials."""
    rng = np.random.default_rng(20260928)
    bits = rng.integers(0, 2 ** 32, size=1 << 23, dtype=np.uint64).astype(np.uint32).view(np.float32)
    dense = np.linspace(-100.0, 100.0, 1 << 22, dtype=np.float64).astype(np.float32)
    near = (rng.standard_normal(1 << 20) * 3).astype(np.float32)
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 88.3762626647949, -88.3762626647949, 88.4, -88.4, 89.0, -104.0,
                   1.17549435e-38, 1e-45, -1e-45, 1.4e-45, 3.4e38, -3.4e38, 0.70710678, 1.0, 0.5, 2.0], dtype=np.float32)
    x = np.concatenate([bits, dense, near, sp])
    if lo is not None:
        x = x[(x >= lo) & (x <= hi)]
    pad = (-x.size) % 4
    return np.concatenate([x, np.zeros(pad, np.float32)])


def _same_bits(a, b):
    """bit-identical, any NaN matching any NaN (payload/sign of a NaN is not a result)"""
    ua, ub = a.view(np.uint32), b.view(np.uint32)
    bad = (ua != ub) & ~(np.isnan(a) & np.isnan(b))
    return int(bad.sum()), bad


@pytest.mark.parametrize("kind,ref_name", [(0, "ref_expfv"), (1, "ref_logfv"), (2, "ref_logisticfv"), (3, "ref_tanhfv"),
                                           (4, "ref_elufv")])
def test_vector_math_bit_for_bit(kind, ref_name):
    R = _load("libflappie_inlref.so")
    x = _sweep()
    assert x.size > 10_000_000
    ref = np.empty_like(x)
    ours = np.empty_like(x)
    getattr(R, ref_name).argtypes = [P(C.c_float), P(C.c_float), C.c_size_t]
    getattr(R, ref_name)(_f(x), _f(ref), x.size // 4)
    L = ffo.lib()
    L.fo_map_array.argtypes = [C.c_int, P(C.c_float), P(C.c_float), C.c_size_t]
    assert L.fo_map_array(kind, _f(x), _f(ours), x.size) == 0
    if kind == 4:
        # elufv returns the whole vector untouched when no lane has its sign bit set (util.h:340-343) and otherwise
        # exp(x)-1 for lanes that compare < 0: a NaN lane takes the exp branch either way -> compare non-NaN lanes
        keep = ~np.isnan(x)
        nbad, bad = _same_bits(ref[keep], ours[keep])
    else:
        nbad, bad = _same_bits(ref, ours)
    assert nbad == 0, "%d of %d differ, first x=%r" % (nbad, x.size, x[np.flatnonzero(bad)[:4]] if kind != 4 else None)


def test_logsumexp_and_phred_bit_for_bit():
    R = _load("libflappie_inlref.so")
    L = ffo.lib()
    rng = np.random.default_rng(7)
    n = 1 << 23
    x = (rng.standard_normal(n) * rng.choice([0.01, 1.0, 10.0, 100.0], n)).astype(np.float32)
    y = (x + (rng.standard_normal(n) * rng.choice([1e-6, 0.1, 1.0, 30.0, 200.0], n))).astype(np.float32)
    x[:8] = [0, -np.inf, np.inf, np.nan, 1, -np.inf, 3e38, -3e38]
    y[:8] = [0, -np.inf, 1, 1, np.nan, 5, 3e38, 3e38]
    a, b = np.empty_like(x), np.empty_like(x)
    R.ref_logsumexpf.argtypes = [P(C.c_float)] * 3 + [C.c_size_t]
    L.fo_logsumexpf_array.argtypes = [P(C.c_float)] * 3 + [C.c_size_t]
    R.ref_logsumexpf(_f(x), _f(y), _f(a), n)
    L.fo_logsumexpf_array(_f(x), _f(y), _f(b), n)
    assert _same_bits(a, b)[0] == 0

    xd, yd = x[: 1 << 21].astype(np.float64) * 1.000001, y[: 1 << 21].astype(np.float64)
    ad, bd = np.empty_like(xd), np.empty_like(xd)
    pd = P(C.c_double)
    R.ref_logsumexp.argtypes = [pd, pd, pd, C.c_size_t]
    L.fo_logsumexp_array.argtypes = [pd, pd, pd, C.c_size_t]
    R.ref_logsumexp(xd.ctypes.data_as(pd), yd.ctypes.data_as(pd), ad.ctypes.data_as(pd), xd.size)
    L.fo_logsumexp_array(xd.ctypes.data_as(pd), yd.ctypes.data_as(pd), bd.ctypes.data_as(pd), xd.size)
    ok = (ad.view(np.uint64) == bd.view(np.uint64)) | (np.isnan(ad) & np.isnan(bd))
    assert ok.all()

    # phredf (util.h:300-305) over probabilities: uniform, close to 1 (the 0.99999 clip), tiny
    p = np.concatenate([rng.random(1 << 22), 1.0 - rng.random(1 << 21) * 1e-4, rng.random(1 << 20) * 1e-6,
                        [0.0, 1.0, 0.99999, 0.999989, 0.9999901]]).astype(np.float32)
    qa, qb = np.empty(p.size, np.int8), np.empty(p.size, np.int8)
    R.ref_phredf.argtypes = [P(C.c_float), C.c_void_p, C.c_size_t]
    L.fo_phredf_array.argtypes = [P(C.c_float), C.c_void_p, C.c_size_t]
    R.ref_phredf(_f(p), qa.ctypes.data, p.size)
    L.fo_phredf_array(_f(p), qb.ctypes.data, p.size)
    assert np.array_equal(qa, qb)
    assert qa.min() >= 33 and qa.max() <= 126


# ---- the reference's compiled decode.c ------------------------------------------------------------------------------
def _decref():
    R = _load("libflappie_decref.so")
    M = P(ffo.FoMat)                      # _Mat and fo_mat share their layout (four size_t + data pointer)
    R.make_flappie_matrix.restype = M
    R.make_flappie_matrix.argtypes = [C.c_size_t, C.c_size_t]
    R.free_flappie_matrix.restype = M
    R.free_flappie_matrix.argtypes = [M]
    R.free_flappie_imatrix.restype = P(ffo.FoIMat)
    R.free_flappie_imatrix.argtypes = [P(ffo.FoIMat)]
    R.decode_crf_flipflop.restype = C.c_float
    R.decode_crf_flipflop.argtypes = [M, C.c_bool, P(C.c_int), P(C.c_float)]
    R.transpost_crf_flipflop.restype = M
    R.transpost_crf_flipflop.argtypes = [M, C.c_bool]
    R.trace_from_posterior.restype = P(ffo.FoIMat)
    R.trace_from_posterior.argtypes = [M]
    R.change_positions.restype = C.c_size_t
    R.change_positions.argtypes = [P(C.c_int), C.c_size_t, P(C.c_int)]
    R.decode_crf_runlength.restype = C.c_float
    R.decode_crf_runlength.argtypes = [M, P(C.c_int)]
    R.transpost_crf_runlength.restype = M
    R.transpost_crf_runlength.argtypes = [M]
    R.exp_activation_inplace.argtypes = [M]
    R.row_normalise_inplace.argtypes = [M]
    return R


def _scores(rng, nparam, nblock, style):
    if style == "normal":
        s = rng.standard_normal((nblock, nparam)) * 2
    elif style == "tanh5":                       # what globalnorm_flipflop emits: 5 tanh(.) - logZ
        s = 5 * np.tanh(rng.standard_normal((nblock, nparam)) * 2) - 3.0
    elif style == "ties":                        # small integers: ties everywhere, exercises every tie rule
        s = rng.integers(-2, 3, (nblock, nparam)).astype(np.float64)
    elif style == "flat":
        s = np.zeros((nblock, nparam))
    elif style == "nan":                         # every score NaN: strict-> scans keep their first candidate
        s = np.full((nblock, nparam), np.nan)
    elif style == "some_nan":
        s = rng.standard_normal((nblock, nparam)) * 2
        s[rng.random((nblock, nparam)) < 0.05] = np.nan
    else:
        raise ValueError(style)
    return s.astype(np.float32)


CASES = [(nb, nblock, style) for nb in (4, 5) for nblock in (1, 2, 3, 17, 800) for style in ("normal", "tanh5", "ties", "flat", "nan", "some_nan")]


@pytest.mark.parametrize("nbase,nblock,style", CASES)
def test_flipflop_decoders_match_compiled_reference(nbase, nblock, style):
    R = _decref()
    L = ffo.lib()
    nstate = 2 * nbase
    nparam = nstate * (nbase + 1)
    rng = np.random.default_rng(1000 * nbase + nblock)
    for rep in range(3):
        dense = _scores(rng, nparam, nblock, style)
        hm = ffo.HostMat.from_dense(dense)          # [nparam x nblock] image
        # Viterbi: path, qpath (qpath[0] = NAN), score; both combine_stays settings
        for combine in (False, True):
            pa, pb = np.full(nblock + 1, -7, np.int32), np.full(nblock + 1, -7, np.int32)
            qa, qb = np.zeros(nblock + 1, np.float32), np.zeros(nblock + 1, np.float32)
            sa = R.decode_crf_flipflop(hm.ptr, combine, pa.ctypes.data_as(P(C.c_int)), _f(qa))
            sb = L.fo_decode_viterbi(hm.ptr, int(combine), pb.ctypes.data_as(P(C.c_int)), _f(qb))
            assert np.float32(sa).view(np.uint32) == np.float32(sb).view(np.uint32)
            assert np.array_equal(pa, pb)
            assert _same_bits(qa, qb)[0] == 0
            # change_positions on that path (decode.c:66-79)
            ca, cb = np.zeros(nblock + 1, np.int32), np.zeros(nblock + 1, np.int32)
            na = R.change_positions(pa.ctypes.data_as(P(C.c_int)), nblock, ca.ctypes.data_as(P(C.c_int)))
            nb_ = L.fo_change_positions(pb.ctypes.data_as(P(C.c_int)), nblock, cb.ctypes.data_as(P(C.c_int)))
            assert na == nb_ and np.array_equal(ca[:na], cb[:nb_])
        # posterior, log and probability space
        for return_log in (True, False):
            ra = R.transpost_crf_flipflop(hm.ptr, return_log)
            rb = L.fo_transpost(hm.ptr, int(return_log))
            a = ffo.take(ra, free=False)
            b = ffo.take(rb, free=False)
            assert _same_bits(a, b)[0] == 0
            if return_log:
                # flappie.c:299-300: exp, then the trace
                R.exp_activation_inplace(ra)
                L.fo_exp_inplace(rb)
                assert _same_bits(ffo.take(ra, free=False), ffo.take(rb, free=False))[0] == 0
                ta = R.trace_from_posterior(ra)
                tb = L.fo_trace_from_posterior(rb)
                assert np.array_equal(ffo.take_i(ta), ffo.take_i(tb))
            R.free_flappie_matrix(ra)
            L.fo_free_mat(rb)


@pytest.mark.parametrize("nblock,style", [(1, "normal"), (5, "ties"), (64, "normal"), (800, "tanh5"), (300, "ties")])
def test_runlength_decoders_match_compiled_reference(nblock, style):
    R = _decref()
    L = ffo.lib()
    nbase = 4
    nparam = 2 * nbase + 2 * nbase * nbase     # shape + scale rows, then the CRF transitions (layers.c:1235-1246)
    rng = np.random.default_rng(nblock)
    dense = _scores(rng, nparam, nblock, style)
    dense[:, : 2 * nbase] = np.abs(dense[:, : 2 * nbase]) + 1.0
    hm = ffo.HostMat.from_dense(dense)
    pa, pb = np.full(nblock + 1, -7, np.int32), np.full(nblock + 1, -7, np.int32)
    sa = R.decode_crf_runlength(hm.ptr, pa.ctypes.data_as(P(C.c_int)))
    sb = L.fo_decode_crf_runlength(hm.ptr, pb.ctypes.data_as(P(C.c_int)))
    assert np.float32(sa).view(np.uint32) == np.float32(sb).view(np.uint32)
    assert np.array_equal(pa, pb)
    ra = R.transpost_crf_runlength(hm.ptr)
    rb = L.fo_transpost_crf_runlength(hm.ptr)
    assert _same_bits(ffo.take(ra, free=False), ffo.take(rb, free=False))[0] == 0
    R.free_flappie_matrix(ra)
    L.fo_free_mat(rb)


def _v1_param(rng, nblock, nbase, style):
    """[nblock x 4 nbase]: shape and scale rows of a discrete Weibull (positive), then move and stay weights"""
    dense = _scores(rng, 4 * nbase, nblock, style)
    dense[:, :nbase] = 0.5 + 2.5 * rng.random((nblock, nbase))          # shape
    dense[:, nbase: 2 * nbase] = 0.3 + 8.0 * rng.random((nblock, nbase))      # scale
    return dense.astype(np.float32)


@pytest.mark.parametrize("nbase,nblock,style", [(4, 1, "normal"), (4, 2, "ties"), (4, 17, "ties"), (4, 64, "normal"), (4, 800, "tanh5"), (4, 300, "flat"),
                                                (4, 500, "ties"), (5, 200, "normal"), (2, 50, "normal"), (4, 40, "some_nan")])
def test_first_generation_runlength_decoders_match_compiled_reference(nbase, nblock, style):
    """decode.h's last five prototypes (decode_runlength, posterior_runlength, runlengths_mean, runlengths_unit, runlength_to_basecall;
    decode.c:552-892): the oracle's restatement against the reference's own object code -- paths, run lengths and strings identical, scores
    and posteriors bit for bit"""
    R = _decref()
    L = ffo.lib()
    M = P(ffo.FoMat)
    ip = P(C.c_int)
    R.decode_runlength.restype = C.c_float
    R.decode_runlength.argtypes = [M, ip]
    R.posterior_runlength.restype = M
    R.posterior_runlength.argtypes = [M]
    for fn in (R.runlengths_mean, R.runlengths_unit):
        fn.restype = C.c_size_t
        fn.argtypes = [M, ip, ip]
    R.runlength_to_basecall.restype = C.c_void_p
    R.runlength_to_basecall.argtypes = [ip, ip, C.c_size_t]
    R.dwmean.restype = C.c_float
    R.dwmean.argtypes = [C.c_float, C.c_float, C.c_int]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(77 * nbase + nblock)
    for rep in range(3):
        dense = _v1_param(rng, nblock, nbase, style)
        hm = ffo.HostMat.from_dense(dense)
        pa, pb = np.full(nblock, -7, np.int32), np.full(nblock, -7, np.int32)
        sa = R.decode_runlength(hm.ptr, pa.ctypes.data_as(ip))
        sb = L.fo_decode_runlength(hm.ptr, pb.ctypes.data_as(ip))
        assert np.float32(sa).view(np.uint32) == np.float32(sb).view(np.uint32)
        assert np.array_equal(pa, pb) and pa.min() >= -1 and pa.max() < nbase
        for ref, mine in ((R.runlengths_mean, L.fo_runlengths_mean), (R.runlengths_unit, L.fo_runlengths_unit)):
            ra, rb = np.full(nblock, -7, np.int32), np.full(nblock, -7, np.int32)
            na = ref(hm.ptr, pa.ctypes.data_as(ip), ra.ctypes.data_as(ip))
            nb_ = mine(hm.ptr, pb.ctypes.data_as(ip), rb.ctypes.data_as(ip))
            assert na == nb_ == int(ra.sum()) and np.array_equal(ra, rb)
            qa = R.runlength_to_basecall(pa.ctypes.data_as(ip), ra.ctypes.data_as(ip), nblock)
            qb = L.fo_runlength_to_basecall(pb.ctypes.data_as(ip), rb.ctypes.data_as(ip), nblock)
            assert C.string_at(qa) == C.string_at(qb) and len(C.string_at(qa)) == na
            libc.free(qa)
            libc.free(qb)
        if style != "some_nan":      # (a NaN weight makes every later vector NaN in both; the bit patterns of those NaNs are libm's business)
            ra = R.posterior_runlength(hm.ptr)
            rb = L.fo_posterior_runlength(hm.ptr)
            a, b = ffo.take(ra, free=False), ffo.take(rb, free=False)
            assert a.shape == (nblock + 1, 4 * nbase) and _same_bits(a, b)[0] == 0
            assert not a[:, : 2 * nbase].any() and not a[nblock].any()
            R.free_flappie_matrix(ra)
            L.fo_free_mat(rb)
    for shape, scale in ((0.5, 0.3), (1.0, 1.0), (2.2, 7.5), (3.0, 0.01), (0.7, 50.0)):
        assert np.float32(R.dwmean(shape, scale, 100)).view(np.uint32) == np.float32(L.fo_dwmean(shape, scale, 100)).view(np.uint32)


def test_glue_row_normalise_known_answers():
    """The glue's row_normalise_inplace (the one arithmetic loop of ref_decode_glue.c that is not the reference's own
    inline code) against the definition, shapes with 0..3 pad lanes; columns then sum to one (test_flappie_matrix.c:32-46)."""
    R = _decref()
    L = ffo.lib()
    rng = np.random.default_rng(3)
    for nr in (1, 2, 3, 4, 5, 7, 40, 60):
        dense = (rng.random((9, nr)) + 0.1).astype(np.float32)
        a = ffo.HostMat.from_dense(dense)
        b = ffo.HostMat.from_dense(dense)
        R.row_normalise_inplace(a.ptr)
        L.fo_row_normalise_inplace(b.ptr)
        assert _same_bits(a.data, b.data)[0] == 0
        img = a.data.reshape(9, -1)[:, :nr]
        assert np.abs(img.sum(axis=1) - 1.0).max() < 1e-6


class _RawTable(C.Structure):
    _fields_ = [("uuid", C.c_char_p), ("n", C.c_size_t), ("start", C.c_size_t), ("end", C.c_size_t), ("raw", P(C.c_float))]


def test_features_from_raw_matches_compiled_reference():
    """nnfeatures.c:15-28 (row A2): raw[start..end) as a [1 x T] matrix, one sample per 4-float column; NULL for n == 0 / raw == NULL"""
    R = _decref()
    L = ffo.lib()
    R.features_from_raw.restype = P(ffo.FoMat)
    R.features_from_raw.argtypes = [_RawTable]
    L.fo_features_from_raw.restype = P(ffo.FoMat)
    L.fo_features_from_raw.argtypes = [P(C.c_float), C.c_size_t, C.c_size_t]
    raw = np.random.default_rng(0).standard_normal(5000).astype(np.float32)
    for start, end in ((0, 5000), (200, 4990), (17, 18)):
        a = R.features_from_raw(_RawTable(None, raw.size, start, end, _f(raw)))
        b = L.fo_features_from_raw(_f(raw), start, end)
        assert a.contents.nr == 1 and a.contents.nc == end - start and a.contents.stride == 4
        A = np.ctypeslib.as_array(a.contents.f, shape=(end - start, 4)).copy()
        B = np.ctypeslib.as_array(b.contents.f, shape=(end - start, 4)).copy()
        assert np.array_equal(A, B) and np.array_equal(A[:, 0], raw[start:end]) and not A[:, 1:].any()
        R.free_flappie_matrix(a)
        L.fo_free_mat(b)
    assert not R.features_from_raw(_RawTable(None, 0, 0, 0, _f(raw)))
    assert not R.features_from_raw(_RawTable(None, 10, 0, 10, None))
