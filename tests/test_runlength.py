"""Run-length ("runnie", model rle_r941_native) path -- SURVEY.md section 8f row N4: LSTM5 trunk with the
globalnorm_runlengthV2 head (layers.c:1325-1358), transpost_crf_runlength and decode_crf_runlength (decode.c:927-1159),
the (base, shape, scale, dwell) records of runnie.c:282-313.  The reference has no test or vector for any of it, so the
oracle's restatement is pinned only by brute force (CPU tests below); the GPU path is then checked against the oracle."""
import ctypes as C
import itertools
import math

import numpy as np
import pytest

from flappie_amd import model as M
from oracle import ffo


def rle_idx(base_from, stay_from, base_to, nbase):
    return base_to * 2 * nbase + base_from + (nbase if stay_from else 0)


def allowed(prev, cur, nbase):
    """state = base (+ nbase if stay).  Into a move state from any state of a different base; into the stay state of a
    base from its own move or stay state (decode.c:1014-1036)."""
    pb, cb = prev % nbase, cur % nbase
    return (pb != cb) if cur < nbase else (pb == cb)


def brute_force(param, nbase):
    """log partition function and best path by enumeration over all state paths (initial vector zero)."""
    T = param.shape[0]
    ns = 2 * nbase
    S = param[:, ns:].astype(np.float64)
    logz, best, best_path = -math.inf, -math.inf, None
    for start in range(ns):
        for path in itertools.product(range(ns), repeat=T):
            prev, sc, ok = start, 0.0, True
            for t, cur in enumerate(path):
                if not allowed(prev, cur, nbase):
                    ok = False
                    break
                sc += S[t, rle_idx(prev % nbase, prev >= nbase, cur % nbase, nbase)]
                prev = cur
            if not ok:
                continue
            logz = np.logaddexp(logz, sc)
            if sc > best:
                best, best_path = sc, path
    return logz, best, best_path


def random_param(rng, T, nbase, scale=1.0):
    P = 2 * nbase * (nbase + 1)
    x = (rng.standard_normal((T, P)) * scale).astype(np.float32)
    x[:, :2 * nbase] = np.abs(x[:, :2 * nbase]) + 1.0
    return x


# ------------------------------------------------------------------------------------ CPU: the oracle itself
def test_oracle_runlength_against_brute_force():
    lib = ffo.lib()
    rng = np.random.default_rng(2)
    for nbase, T in ((2, 4), (2, 5), (3, 3)):
        param = random_param(rng, T, nbase)
        logz, best, best_path = brute_force(param, nbase)
        pm = ffo.HostMat.from_dense(param)
        assert abs(lib.fo_runlengthV2_partition_function(pm.ptr) - logz) <= 2e-6 * max(1.0, abs(logz))   # stays go through float logsumexpf
        path = np.zeros(T, dtype=np.int32)
        score = lib.fo_decode_crf_runlength(pm.ptr, path.ctypes.data_as(C.POINTER(C.c_int)))
        assert abs(score - best) <= 1e-5 and tuple(path) == best_path
        # transition posteriors: exp(post) summed over a block's transitions is the same for every block (= Z)
        post = ffo.take(lib.fo_transpost_crf_runlength(pm.ptr))
        np.testing.assert_array_equal(post[:, :2 * nbase], param[:, :2 * nbase])
        # every (from-state, to-base) pair is exactly one allowed transition, so the rows 2*nbase.. cover them all
        mass = [np.logaddexp.reduce(post[t, 2 * nbase:].astype(np.float64)) for t in range(T)]
        assert np.ptp(mass) <= 1e-4 and abs(mass[0] - logz) <= 1e-4


def test_oracle_head_and_records():
    lib = ffo.lib()
    for x in (-30.0, -1.0, 0.0, 0.5, 20.0):
        assert abs(lib.fo_softplusf(x) - math.log1p(math.exp(x))) <= 1e-6 * max(1.0, abs(x))
    rng = np.random.default_rng(4)
    nbase, H, T = 4, 24, 30
    P = 2 * nbase * (nbase + 1)
    x = np.tanh(rng.standard_normal((T, H))).astype(np.float32)
    w = (rng.standard_normal((P, H)) / np.sqrt(H) * 2).astype(np.float32)
    b = rng.standard_normal((1, P)).astype(np.float32)
    for temperature in (1.0, 0.6):
        got = ffo.take(lib.fo_globalnorm_runlengthV2(ffo.HostMat.from_dense(x).ptr, ffo.HostMat.from_dense(w).ptr,
                                                     ffo.HostMat.from_dense(b).ptr, temperature))
        a = x.astype(np.float64) @ w.T.astype(np.float64) + b
        np.testing.assert_allclose(got[:, :nbase], 1 + np.log1p(np.exp(a[:, :nbase])), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(got[:, nbase:2 * nbase], 1e-8 + np.log1p(np.exp(a[:, nbase:2 * nbase])), rtol=2e-6, atol=2e-6)
        # globally normalised: the partition function of the result is ~0
        z = lib.fo_runlengthV2_partition_function(ffo.HostMat.from_dense(got).ptr)
        assert abs(z) <= 1e-3
        raw = 5 * np.tanh(a[:, 2 * nbase:]) / temperature
        shift = raw - got[:, 2 * nbase:]
        assert np.ptp(shift) <= 1e-4                                  # one constant (logZ / nblock) subtracted everywhere
    path = np.array([1, 5, 5, 2, 0, 4, 4, 4, 3], dtype=np.int32)     # C stay stay G A stay stay stay T
    ip = C.POINTER(C.c_int)
    base, block, dwell = (np.zeros(9, dtype=np.int32) for _ in range(3))
    n = lib.fo_runlength_records(path.ctypes.data_as(ip), 9, 4, base.ctypes.data_as(ip), block.ctypes.data_as(ip), dwell.ctypes.data_as(ip))
    assert n == 4 and list(base[:4]) == [1, 2, 0, 3] and list(block[:4]) == [0, 3, 4, 8] and list(dwell[:4]) == [3, 1, 4, 1]


# ------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def B():
    from flappie_amd import binding
    return binding


@pytest.fixture(scope="module")
def engine(B):
    e = B.Engine(0)
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,T,nread", [(64, 2000, 5), (96, 1237, 3), (48, 400, 18)])
def test_runlength_pipeline_matches_oracle(B, engine, hidden, T, nread):
    mdl = M.synthetic_model(M.NET_LSTM5_RLE, hidden, seed=5)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    sig = np.random.default_rng(T).standard_normal((nread, T)).astype(np.float32)
    for viterbi_only, temperature in ((False, 1.0), (True, 0.8)):
        b = B.Batch(dm, nread, T)
        b.set_signals(sig)
        b.run(temperature, B.RUN_VITERBI_ONLY if viterbi_only else 0)
        b.finish()
        for r in range(nread):
            ref = om.runlength_call(sig[r], temperature=temperature, viterbi_only=viterbi_only)
            assert np.abs(b.transitions(r) - ref["param"]).max() <= 1e-4
            path, _ = b.path(r)
            assert np.array_equal(path[:-1], ref["path"]), r                       # bit-exact decode
            # scores and posteriors are UN-normalised sums over blocks: input differences of <= 1e-4 per entry add up
            nblock = ref["param"].shape[0]
            # (the posterior path score sums nblock posteriors that each carry an O(nblock) accumulated difference)
            assert abs(b.score(r) - ref["score"]) <= max(1e-4 * nblock, 1e-3 * abs(ref["score"]))
            if not viterbi_only:
                assert np.abs(b.posterior(r) - ref["post"]).max() <= 1e-4 * nblock
            assert b.basecall(r) == ""                                              # no flip-flop strings for this model
        b.close()
    dm.close()


@pytest.mark.gpu
def test_runlength_host_api(B):
    """The reference-named C entry points: globalnorm_runlengthV2, runlengthV2_partition_function,
    transpost_crf_runlength, decode_crf_runlength."""
    import os
    from test_host_layer import CMat, HOSTLIB, _f
    PM = C.POINTER(CMat)
    L = C.CDLL(HOSTLIB)
    L.mat_from_array.restype = PM
    L.mat_from_array.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t]
    L.globalnorm_runlengthV2.restype = PM
    L.globalnorm_runlengthV2.argtypes = [PM, PM, PM, C.c_float, PM]
    L.runlengthV2_partition_function.restype = C.c_double
    L.runlengthV2_partition_function.argtypes = [PM]
    L.transpost_crf_runlength.restype = PM
    L.transpost_crf_runlength.argtypes = [PM]
    L.decode_crf_runlength.restype = C.c_float
    L.decode_crf_runlength.argtypes = [PM, C.POINTER(C.c_int)]
    L.nbase_from_crf_runlength_nparam.restype = C.c_size_t
    L.nbase_from_crf_runlength_nparam.argtypes = [C.c_size_t]

    def mk(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return L.mat_from_array(_f(a), a.shape[1], a.shape[0])

    def dense(pm):
        m = pm.contents
        return np.ctypeslib.as_array(m.f, shape=(m.nc, m.stride))[:, : m.nr].copy()

    lib = ffo.lib()
    rng = np.random.default_rng(8)
    nbase, H, T = 4, 96, 150
    P = 40
    assert L.nbase_from_crf_runlength_nparam(P) == 4
    x = np.tanh(rng.standard_normal((T, H))).astype(np.float32)
    w = (rng.standard_normal((P, H)) / np.sqrt(H) * 3).astype(np.float32)
    bias = rng.standard_normal((1, P)).astype(np.float32)
    want = ffo.take(lib.fo_globalnorm_runlengthV2(ffo.HostMat.from_dense(x).ptr, ffo.HostMat.from_dense(w).ptr, ffo.HostMat.from_dense(bias).ptr, 0.9))
    c = L.globalnorm_runlengthV2(mk(x), mk(w), mk(bias), 0.9, None)
    assert c
    np.testing.assert_allclose(dense(c), want, rtol=0, atol=5e-5)
    param = random_param(rng, T, nbase, scale=2.0)
    pm = ffo.HostMat.from_dense(param)
    z = L.runlengthV2_partition_function(mk(param))
    zw = lib.fo_runlengthV2_partition_function(pm.ptr)
    assert abs(z - zw) <= 1e-9 * max(1.0, abs(zw))          # fp64 re-association only (the stay updates are the same float logsumexpf)
    post = L.transpost_crf_runlength(mk(param))
    np.testing.assert_allclose(dense(post), ffo.take(lib.fo_transpost_crf_runlength(pm.ptr)), rtol=0, atol=2e-4)
    path, ref_path = np.zeros(T, dtype=np.int32), np.zeros(T, dtype=np.int32)
    ip = C.POINTER(C.c_int)
    s = L.decode_crf_runlength(mk(param), path.ctypes.data_as(ip))
    sw = lib.fo_decode_crf_runlength(pm.ptr, ref_path.ctypes.data_as(ip))
    assert np.array_equal(path, ref_path) and abs(s - sw) <= 1e-3
    assert np.isnan(L.decode_crf_runlength(None, path.ctypes.data_as(ip))) and not L.transpost_crf_runlength(None)


@pytest.mark.gpu
@pytest.mark.parametrize("nbase,T,style", [(4, 1, "normal"), (4, 3, "ties"), (4, 150, "normal"), (4, 777, "ties"), (4, 5000, "tanh"), (5, 300, "normal"), (4, 9000, "flat")])
def test_first_generation_decoders_through_the_c_api(B, nbase, T, style):
    """decode_runlength, posterior_runlength, runlengths_mean, runlengths_unit, runlength_to_basecall (decode.c:552-892; k_rl1_viterbi,
    k_rl1_posterior, k_rl1_mean) against the oracle, which tests/test_ref_pins.py holds to the reference's object code: paths, run lengths and
    strings identical, the Viterbi score bit for bit (additions and comparisons only), the posterior within the logsumexp tolerance of the
    other decoders; 5000 and 9000 blocks cross the kernels' 4096-block traceback chunks"""
    from test_host_layer import CMat, HOSTLIB, _f
    PM = C.POINTER(CMat)
    ip = C.POINTER(C.c_int)
    L = C.CDLL(HOSTLIB)
    L.mat_from_array.restype = PM
    L.mat_from_array.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t]
    L.decode_runlength.restype = C.c_float
    L.decode_runlength.argtypes = [PM, ip]
    L.posterior_runlength.restype = PM
    L.posterior_runlength.argtypes = [PM]
    for fn in (L.runlengths_mean, L.runlengths_unit):
        fn.restype = C.c_size_t
        fn.argtypes = [PM, ip, ip]
    L.runlength_to_basecall.restype = C.c_void_p
    L.runlength_to_basecall.argtypes = [ip, ip, C.c_size_t]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    lib = ffo.lib()
    rng = np.random.default_rng(1000 * nbase + T)
    if style == "ties":
        w = rng.integers(-2, 3, (T, 2 * nbase)).astype(np.float32)
    elif style == "flat":
        w = np.zeros((T, 2 * nbase), dtype=np.float32)
    elif style == "tanh":
        w = (5 * np.tanh(rng.standard_normal((T, 2 * nbase)) * 2) - 1.5).astype(np.float32)
    else:
        w = (rng.standard_normal((T, 2 * nbase)) * 2).astype(np.float32)
    param = np.concatenate([(0.5 + 2.5 * rng.random((T, nbase))).astype(np.float32), (0.3 + 8.0 * rng.random((T, nbase))).astype(np.float32), w], axis=1)
    a = np.ascontiguousarray(param, dtype=np.float32)
    cm = L.mat_from_array(_f(a), a.shape[1], a.shape[0])
    pm = ffo.HostMat.from_dense(param)
    path, want = np.full(T, -7, np.int32), np.full(T, -7, np.int32)
    s = L.decode_runlength(cm, path.ctypes.data_as(ip))
    sw = lib.fo_decode_runlength(pm.ptr, want.ctypes.data_as(ip))
    assert np.array_equal(path, want)
    assert np.float32(s).view(np.uint32) == np.float32(sw).view(np.uint32)
    for mine, ref in ((L.runlengths_mean, lib.fo_runlengths_mean), (L.runlengths_unit, lib.fo_runlengths_unit)):
        rl, rw = np.full(T, -7, np.int32), np.full(T, -7, np.int32)
        n = mine(cm, path.ctypes.data_as(ip), rl.ctypes.data_as(ip))
        nw = ref(pm.ptr, want.ctypes.data_as(ip), rw.ctypes.data_as(ip))
        assert n == nw and np.array_equal(rl, rw)
        q = L.runlength_to_basecall(path.ctypes.data_as(ip), rl.ctypes.data_as(ip), T)
        qw = lib.fo_runlength_to_basecall(want.ctypes.data_as(ip), rw.ctypes.data_as(ip), T)
        assert C.string_at(q) == C.string_at(qw) and len(C.string_at(q)) == n
        libc.free(q)
        libc.free(qw)
    post = L.posterior_runlength(cm)
    m = post.contents
    got = np.ctypeslib.as_array(m.f, shape=(m.nc, m.stride))[:, : m.nr].copy()
    ref = ffo.take(lib.fo_posterior_runlength(pm.ptr))
    assert got.shape == ref.shape == (T + 1, 4 * nbase)
    assert np.abs(got - ref).max() <= 2e-5 + 2e-6 * np.abs(ref).max()
    assert not got[:, : 2 * nbase].any() and not got[T].any()
    assert np.isnan(L.decode_runlength(None, path.ctypes.data_as(ip))) and not L.posterior_runlength(None) and 0 == L.runlengths_mean(None, None, None)


# ------------------------------------------------------------------------------------ the first-generation head (layers.c:1115-1228)
def _brute_force_v1(param, nbase):
    """log-sum over every state path s_-1, s_0 .. s_T-1 of: move[s_t] when the base changes, stay[s_t] when it does not
    (the start state s_-1 is free and costs nothing: runlength_partition_function starts from zeros)"""
    import itertools
    T = param.shape[0]
    move, stay = param[:, 2 * nbase:3 * nbase].astype(np.float64), param[:, 3 * nbase:4 * nbase].astype(np.float64)
    logz = -np.inf
    for path in itertools.product(range(nbase), repeat=T + 1):
        sc = sum(stay[t, path[t + 1]] if path[t + 1] == path[t] else move[t, path[t + 1]] for t in range(T))
        logz = np.logaddexp(logz, sc)
    return logz


def test_oracle_first_generation_head():
    lib = ffo.lib()
    rng = np.random.default_rng(12)
    for nbase, T in ((2, 5), (3, 4), (4, 3)):
        param = rng.standard_normal((T, 4 * nbase)).astype(np.float32)
        want = _brute_force_v1(param, nbase)
        got = lib.fo_runlength_partition_function(ffo.HostMat.from_dense(param).ptr)
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want))
    nbase, H, T = 4, 24, 30
    x = np.tanh(rng.standard_normal((T, H))).astype(np.float32)
    w = (rng.standard_normal((4 * nbase, H)) / np.sqrt(H) * 2).astype(np.float32)
    b = rng.standard_normal((1, 4 * nbase)).astype(np.float32)
    for temperature in (1.0, 0.6):
        got = ffo.take(lib.fo_globalnorm_runlength(ffo.HostMat.from_dense(x).ptr, ffo.HostMat.from_dense(w).ptr, ffo.HostMat.from_dense(b).ptr, temperature))
        a = x.astype(np.float64) @ w.T.astype(np.float64) + b
        np.testing.assert_allclose(got[:, :nbase], 1 + np.log1p(np.exp(a[:, :nbase])), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(got[:, nbase:2 * nbase], 0.1 + np.log1p(np.exp(a[:, nbase:2 * nbase])), rtol=2e-6, atol=2e-6)
        assert abs(lib.fo_runlength_partition_function(ffo.HostMat.from_dense(got).ptr)) <= 1e-3     # globally normalised
        assert np.ptp(5 * np.tanh(a[:, 2 * nbase:]) / temperature - got[:, 2 * nbase:]) <= 1e-4       # one constant subtracted


@pytest.mark.gpu
def test_first_generation_head_host_api(B):
    """globalnorm_runlength, runlength_partition_function, nbase_from_runlength_nparam through the reference-named C calls"""
    from test_host_layer import CMat, HOSTLIB, _f
    PM = C.POINTER(CMat)
    L = C.CDLL(HOSTLIB)
    L.mat_from_array.restype = PM
    L.mat_from_array.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t]
    L.globalnorm_runlength.restype = PM
    L.globalnorm_runlength.argtypes = [PM, PM, PM, C.c_float, PM]
    L.runlength_partition_function.restype = C.c_double
    L.runlength_partition_function.argtypes = [PM]
    L.nbase_from_runlength_nparam.restype = C.c_size_t
    L.nbase_from_runlength_nparam.argtypes = [C.c_size_t]

    def mk(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return L.mat_from_array(_f(a), a.shape[1], a.shape[0])

    def dense(pm):
        m = pm.contents
        return np.ctypeslib.as_array(m.f, shape=(m.nc, m.stride))[:, : m.nr].copy()

    lib = ffo.lib()
    rng = np.random.default_rng(9)
    assert L.nbase_from_runlength_nparam(16) == 4
    for nbase, H, T in ((4, 96, 150), (5, 64, 33), (4, 256, 1)):
        P = 4 * nbase
        x = np.tanh(rng.standard_normal((T, H))).astype(np.float32)
        w = (rng.standard_normal((P, H)) / np.sqrt(H) * 3).astype(np.float32)
        bias = rng.standard_normal((1, P)).astype(np.float32)
        want = ffo.take(lib.fo_globalnorm_runlength(ffo.HostMat.from_dense(x).ptr, ffo.HostMat.from_dense(w).ptr, ffo.HostMat.from_dense(bias).ptr, 0.9))
        c = L.globalnorm_runlength(mk(x), mk(w), mk(bias), 0.9, None)
        assert c
        np.testing.assert_allclose(dense(c), want, rtol=0, atol=5e-5)
        param = (rng.standard_normal((T, P)) * 2).astype(np.float32)
        z = L.runlength_partition_function(mk(param))
        zw = lib.fo_runlength_partition_function(ffo.HostMat.from_dense(param).ptr)
        assert abs(z - zw) <= 1e-9 * max(1.0, abs(zw))
    assert not L.globalnorm_runlength(None, None, None, 1.0, None)
