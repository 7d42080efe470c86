"""GPU decode operators through the reference-named C functions (include/decode.h -> libflappie_host.so -> ffhip_viterbi /
ffhip_transpost / ffhip_trace) on score matrices the network would never emit but a caller of decode.h may pass: ties
everywhere, constant scores, NaNs.  The oracle these are held to is itself bit-for-bit the reference's compiled decode.c on
the same inputs (tests/test_ref_pins.py).

Integer results (path, change positions) must be EQUAL.  Float results: Viterbi qpath/score are sums of the input scores along
the path -> equal bits; posteriors within 2e-5 + 2e-6 |x| (log-sum-exp association differs), trace within one count, with the
measured fraction of differing trace cells printed (run pytest -s / -rP to see it)."""
import ctypes as C

import numpy as np
import pytest

from test_host_layer import CIMat, CMat, _dense, _f, host  # noqa: F401  (host is a fixture)

pytestmark = pytest.mark.gpu
P = C.POINTER


def _scores(rng, nparam, nblock, style):
    if style == "normal":
        s = rng.standard_normal((nblock, nparam)) * 2
    elif style == "tanh5":
        s = 5 * np.tanh(rng.standard_normal((nblock, nparam)) * 2) - 3.0
    elif style == "ties":
        s = rng.integers(-2, 3, (nblock, nparam)).astype(np.float64)
    elif style == "flat":
        s = np.zeros((nblock, nparam))
    elif style == "nan":
        s = np.full((nblock, nparam), np.nan)
    elif style == "some_nan":
        s = rng.standard_normal((nblock, nparam)) * 2
        s[rng.random((nblock, nparam)) < 0.05] = np.nan
    elif style == "rare_bad":                    # a NaN or an infinity every few thousand entries: most groups of 8 blocks take the fast
        s = rng.standard_normal((nblock, nparam)) * 2                 # chain of k_viterbi8x / 10x, a few the literal scan, back and forth
        u = rng.random((nblock, nparam))
        s[u < 1e-4] = np.nan
        s[(u >= 1e-4) & (u < 2e-4)] = np.inf
        s[(u >= 2e-4) & (u < 3e-4)] = -np.inf
    elif style == "wide":                        # block ranges beyond kFbRange: these reads keep the log-space posterior kernels
        s = rng.standard_normal((nblock, nparam)) * 60
    elif style == "nan_block":                   # one all-NaN block in the middle of an ordinary read
        s = rng.standard_normal((nblock, nparam)) * 2
        s[nblock // 2] = np.nan
    return s.astype(np.float32)


def _same_bits(a, b):
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("nbase", [4, 5])
@pytest.mark.parametrize("style", ["normal", "tanh5", "ties", "flat", "nan", "some_nan", "nan_block", "rare_bad"])
def test_viterbi_any_scores_equal_the_oracle(host, nbase, style):
    """decode_crf_flipflop (decode.c:119-204) incl. the cases ADVICE r1 named: an all-NaN matrix must give state 0 everywhere
    (the reference's strict-> scans keep their first candidate), never an out-of-range state or an out-of-bounds read."""
    from oracle import ffo
    L = ffo.lib()
    nstate = 2 * nbase
    nparam = nstate * (nbase + 1)
    rng = np.random.default_rng(nbase * 100 + len(style))
    try:
        for nblock in (1, 7, 8, 9, 17, 300, 2500, 4500):         # 2500, 4500 > one / two traceback chunks of the kernels; 8, 9, 17: edges of the groups of 8 blocks
            dense = _scores(rng, nparam, nblock, style)
            m = host.mat_from_array(_f(np.ascontiguousarray(dense)), nparam, nblock)
            hm = ffo.HostMat.from_dense(dense)
            for combine in (False, True):
                pa, pb = np.full(nblock + 1, -7, np.int32), np.full(nblock + 1, -7, np.int32)
                qa, qb = np.zeros(nblock + 1, np.float32), np.zeros(nblock + 1, np.float32)
                sa = host.decode_crf_flipflop(m, combine, pa.ctypes.data_as(P(C.c_int)), _f(qa))
                sb = L.fo_decode_viterbi(hm.ptr, int(combine), pb.ctypes.data_as(P(C.c_int)), _f(qb))
                lo = -1 if combine else 0
                assert pa.min() >= lo and pa.max() < nstate, (style, nblock, pa.min(), pa.max())
                assert np.array_equal(pa, pb), (style, nblock, combine)
                assert _same_bits(qa, qb), (style, nblock)
                assert _same_bits(np.float32([sa]), np.float32([sb])), (style, nblock, sa, sb)
            host.free_flappie_matrix(m)
    finally:
        host.flappie_hip_shutdown()


@pytest.mark.parametrize("nbase", [4, 5])
def test_posterior_and_trace_tolerances_with_hit_rates(host, nbase):
    """transpost_crf_flipflop + exp + trace_from_posterior against the oracle, counting how often the tolerated differences
    actually occur: fraction of trace cells off by one count, largest log-posterior difference."""
    from oracle import ffo
    L = ffo.lib()
    nstate = 2 * nbase
    nparam = nstate * (nbase + 1)
    rng = np.random.default_rng(nbase)
    ncell = noff = 0
    worst = 0.0
    try:
        for nblock in (1, 2, 3, 33, 64, 65, 129, 800, 2000):        # 32: blocks per staged chunk of the chains; 64: per flush of their vectors
            for style in ("tanh5", "normal", "ties") + (("wide",) if nblock in (65, 800) else ()):
                dense = _scores(rng, nparam, nblock, style)
                # globally normalised, as globalnorm_flipflop hands them over (layers.c:1089-1096): without that the forward
                # values grow like nblock * mean score and fp32 cannot hold posteriors to 1e-5 for anybody
                dense = (dense - np.float32(L.fo_partition_function(ffo.HostMat.from_dense(dense).ptr) / nblock)).astype(np.float32)
                m = host.mat_from_array(_f(np.ascontiguousarray(dense)), nparam, nblock)
                hm = ffo.HostMat.from_dense(dense)
                post = host.transpost_crf_flipflop(m, True)
                ref = L.fo_transpost(hm.ptr, 1)
                a, b = _dense(post), ffo.take(ref, free=False)
                # "wide" (scores of +-200, outside any model: the reference's own fp32 forward values are then in the hundreds, their
                # ulp 3e-5): one more term, an ulp at the magnitude of the scores
                extra = 1e-6 * float(np.abs(dense).max()) if style == "wide" else 0.0
                assert np.all(np.abs(a - b) <= 2e-5 + 2e-6 * np.abs(b) + extra), (style, nblock, float(np.abs(a - b).max()))
                if style != "wide":
                    worst = max(worst, float(np.abs(a - b).max()))
                host.exp_activation_inplace(post)
                L.fo_exp_inplace(ref)
                tr = host.trace_from_posterior(post)
                t = np.ctypeslib.as_array(tr.contents.f, shape=(nblock + 1, tr.contents.stride))[:, : tr.contents.nr].copy()
                tref = ffo.take_i(L.fo_trace_from_posterior(ref))
                d = np.abs(t - tref)
                assert d.max() <= 1
                ncell += d.size
                noff += int((d == 1).sum())
                host.free_flappie_imatrix(tr)
                host.free_flappie_matrix(post)
                host.free_flappie_matrix(m)
                L.fo_free_mat(ref)
    finally:
        host.flappie_hip_shutdown()
    print("nbase %d: trace cells off by one count: %d of %d (%.4f %%); largest |dlogpost| %.2e" % (nbase, noff, ncell, 100.0 * noff / ncell, worst))
    assert noff <= 0.01 * ncell          # the +-1 tolerance is a rounding-boundary effect, not a licence


def test_constant_signal_read_in_a_batch(engine):
    """A read whose signal is NaN throughout (what med-MAD normalisation makes of a constant signal: 0/0) next to ordinary
    reads: results of the ordinary reads are unaffected, the NaN read's path stays in range and equals the oracle's (the
    reference's exp_ps clamps NaN away inside the gates, so its scores are finite after the first recurrent layer)."""
    from flappie_amd import binding as B, model as M
    from oracle import ffo
    mdl = M.synthetic_model(M.NET_LSTM5, 64, seed=7)
    om = ffo.OracleModel(mdl)
    rng = np.random.default_rng(4)
    sig = rng.standard_normal((3, 1000)).astype(np.float32)
    sig[1] = np.nan
    dm = B.DeviceModel(engine, mdl)
    b = B.Batch(dm, 3, 1000)
    try:
        b.set_signals(sig)
        b.run()
        b.finish()
        for r in (0, 2):
            ref = om.basecall(sig[r])
            assert b.basecall(r) == ref["basecall"] and b.quality(r) == ref["quality"]
            assert np.abs(b.transitions(r) - ref["trans"]).max() <= 1e-4
        ref = om.basecall(sig[1])
        path, _ = b.path(1)
        assert path.min() >= 0 and path.max() < 8
        tr, rt = b.transitions(1), ref["trans"]
        assert np.array_equal(np.isnan(tr), np.isnan(rt))
        if not np.isnan(rt).any():
            assert np.abs(tr - rt).max() <= 1e-4
            assert np.array_equal(path, ref["path"])
            assert b.basecall(1) == ref["basecall"]
    finally:
        b.close()
        dm.close()


def _decref():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "oracle", "_ref", "libflappie_decref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libflappie_decref.so not built (reference sources absent where build() ran)")
    from oracle import ffo
    R = C.CDLL(path)
    M_ = P(ffo.FoMat)
    R.argmax_decoder.restype = C.c_float
    R.argmax_decoder.argtypes = [M_, P(C.c_int)]
    R.constrained_crf_flipflop.restype = C.c_float
    R.constrained_crf_flipflop.argtypes = [M_, P(C.c_int)]
    R.posterior_crf_flipflop.restype = M_
    R.posterior_crf_flipflop.argtypes = [M_, C.c_bool]
    R.free_flappie_matrix.restype = M_
    R.free_flappie_matrix.argtypes = [M_]
    return R


@pytest.mark.parametrize("nbase", [4, 5])
def test_other_flipflop_decoders_against_the_compiled_reference(host, nbase):
    """argmax_decoder (decode.c:17-36), constrained_crf_flipflop (:209-270), posterior_crf_flipflop (:275-372) -- the entry points of
    decode.h outside flappie.c's path -- against the REFERENCE's own decode.c object code (oracle/_ref/libflappie_decref.so):
    integer paths equal, the argmax score (a sum in block order) equal to the bit, log-sum-exp results within 2e-5 + 2e-6 |x|."""
    from oracle import ffo
    R = _decref()
    host.argmax_decoder.restype = C.c_float
    host.argmax_decoder.argtypes = [P(CMat), P(C.c_int)]
    host.constrained_crf_flipflop.restype = C.c_float
    host.constrained_crf_flipflop.argtypes = [P(CMat), P(C.c_int)]
    host.posterior_crf_flipflop.restype = P(CMat)
    host.posterior_crf_flipflop.argtypes = [P(CMat), C.c_bool]
    L = ffo.lib()
    nstate, nparam = 2 * nbase, 2 * nbase * (nbase + 1)
    rng = np.random.default_rng(17 + nbase)
    try:
        for nblock in (1, 5, 800):
            for style in ("normal", "ties", "tanh5"):
                # per-state scores [nstate x nblock]: argmax decoder and constrained Viterbi
                st = _scores(rng, nstate, nblock, style)
                m = host.mat_from_array(_f(np.ascontiguousarray(st)), nstate, nblock)
                hm = ffo.HostMat.from_dense(st)
                sa, sb = np.zeros(nblock, np.int32), np.zeros(nblock, np.int32)
                va = host.argmax_decoder(m, sa.ctypes.data_as(P(C.c_int)))
                vb = R.argmax_decoder(hm.ptr, sb.ctypes.data_as(P(C.c_int)))
                assert np.array_equal(sa, sb) and _same_bits(np.float32([va]), np.float32([vb])), (style, nblock)
                pa, pb = np.zeros(nblock + 1, np.int32), np.zeros(nblock + 1, np.int32)
                va = host.constrained_crf_flipflop(m, pa.ctypes.data_as(P(C.c_int)))
                vb = R.constrained_crf_flipflop(hm.ptr, pb.ctypes.data_as(P(C.c_int)))
                assert np.array_equal(pa, pb), (style, nblock)
                assert _same_bits(np.float32([va]), np.float32([vb])), (style, nblock, va, vb)
                host.free_flappie_matrix(m)
                # transition scores [nparam x nblock], globally normalised: per-state posteriors
                tr = _scores(rng, nparam, nblock, style)
                tr = (tr - np.float32(L.fo_partition_function(ffo.HostMat.from_dense(tr).ptr) / nblock)).astype(np.float32)
                m = host.mat_from_array(_f(np.ascontiguousarray(tr)), nparam, nblock)
                hm = ffo.HostMat.from_dense(tr)
                for return_log in (True, False):
                    a = host.posterior_crf_flipflop(m, return_log)
                    b = R.posterior_crf_flipflop(hm.ptr, return_log)
                    A, Bm = _dense(a), ffo.take(b, free=False)
                    assert A.shape == Bm.shape == (nblock + 1, nstate)
                    assert np.all(np.abs(A - Bm) <= 2e-5 + 2e-6 * np.abs(Bm)), (style, nblock, return_log, float(np.abs(A - Bm).max()))
                    host.free_flappie_matrix(a)
                    R.free_flappie_matrix(b)
                host.free_flappie_matrix(m)
    finally:
        host.flappie_hip_shutdown()
