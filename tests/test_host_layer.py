"""Tests of the C host layer (libflappie_host.so): the reference-compatible API over the HIP engine.
CPU tests cover what needs no GPU (registry names, .mdl reader, signal preparation, matrix type);
GPU tests call calculate_transitions / transpost / decode / trace exactly as flappie.c:245-316 does."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from flappie_amd import model as M

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOSTLIB = os.path.join(ROOT, "flappie_amd", "libflappie_host.so")


class CMat(C.Structure):
    _fields_ = [("nr", C.c_size_t), ("nrq", C.c_size_t), ("nc", C.c_size_t), ("stride", C.c_size_t),
                ("f", C.POINTER(C.c_float)), ("dev", C.c_void_p), ("dev_state", C.c_int)]


class CIMat(C.Structure):
    _fields_ = [("nr", C.c_size_t), ("nrq", C.c_size_t), ("nc", C.c_size_t), ("stride", C.c_size_t),
                ("f", C.POINTER(C.c_int32))]


class RawTable(C.Structure):
    _fields_ = [("uuid", C.c_char_p), ("n", C.c_size_t), ("start", C.c_size_t), ("end", C.c_size_t),
                ("raw", C.POINTER(C.c_float))]


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(HOSTLIB):
        pytest.fail("libflappie_host.so not built: run __graft_entry__.build()")
    L = C.CDLL(HOSTLIB)
    L.get_flappie_model_type.restype = C.c_int
    L.get_flappie_model_type.argtypes = [C.c_char_p]
    L.flappie_model_string.restype = C.c_char_p
    L.flappie_model_string.argtypes = [C.c_int]
    L.flappie_model_description.restype = C.c_char_p
    L.flappie_model_description.argtypes = [C.c_int]
    L.make_flappie_matrix.restype = C.POINTER(CMat)
    L.make_flappie_matrix.argtypes = [C.c_size_t, C.c_size_t]
    L.remake_flappie_matrix.restype = C.POINTER(CMat)
    L.remake_flappie_matrix.argtypes = [C.POINTER(CMat), C.c_size_t, C.c_size_t]
    L.free_flappie_matrix.restype = C.POINTER(CMat)
    L.free_flappie_matrix.argtypes = [C.POINTER(CMat)]
    L.free_flappie_imatrix.restype = C.POINTER(CIMat)
    L.free_flappie_imatrix.argtypes = [C.POINTER(CIMat)]
    L.mat_from_array.restype = C.POINTER(CMat)
    L.mat_from_array.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t]
    L.medianf.restype = C.c_float
    L.medianf.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    L.medmad_normalise_array.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    L.trim_and_segment_raw.restype = RawTable
    L.trim_and_segment_raw.argtypes = [RawTable, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
    L.trim_raw_by_mad.restype = RawTable
    L.trim_raw_by_mad.argtypes = [RawTable, C.c_size_t, C.c_float]
    L.mdl_load.restype = C.c_void_p
    L.mdl_load.argtypes = [C.c_char_p]
    L.mdl_matrix.restype = C.POINTER(CMat)
    L.mdl_matrix.argtypes = [C.c_void_p, C.c_char_p]
    L.mdl_define.restype = C.c_int
    L.mdl_define.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.mdl_free.argtypes = [C.c_void_p]
    L.calculate_transitions.restype = C.POINTER(CMat)
    L.calculate_transitions.argtypes = [RawTable, C.c_float, C.c_int]
    L.transpost_crf_flipflop.restype = C.POINTER(CMat)
    L.transpost_crf_flipflop.argtypes = [C.POINTER(CMat), C.c_bool]
    L.decode_crf_flipflop.restype = C.c_float
    L.decode_crf_flipflop.argtypes = [C.POINTER(CMat), C.c_bool, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.trace_from_posterior.restype = C.POINTER(CIMat)
    L.trace_from_posterior.argtypes = [C.POINTER(CMat)]
    L.exp_activation_inplace.argtypes = [C.POINTER(CMat)]
    L.change_positions.restype = C.c_size_t
    L.change_positions.argtypes = [C.POINTER(C.c_int), C.c_size_t, C.POINTER(C.c_int)]
    L.flappie_hip_shutdown.restype = None
    L.flappie_matrix_sync.argtypes = [C.POINTER(CMat)]
    L.flappie_matrix_to_device.restype = C.c_bool
    L.flappie_matrix_to_device.argtypes = [C.POINTER(CMat)]
    L.flappie_matrix_host_changed.argtypes = [C.POINTER(CMat)]
    L.ffhip_copy_counts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    L.ffhip_set_matrix_policy.argtypes = [C.c_int]
    L.tanh_activation_inplace.argtypes = [C.POINTER(CMat)]
    L.array_from_flappie_matrix.restype = C.POINTER(C.c_float)
    L.array_from_flappie_matrix.argtypes = [C.POINTER(CMat)]
    global _HOST
    _HOST = L
    return L


_HOST = None


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dense(pm):
    """the meaningful rows of a matrix's HOST image; a matrix whose current image is on the device (dev_state 2) is synchronised first,
    as include/flappie_matrix.h asks of code that reads data.f itself"""
    m = pm.contents
    if m.dev_state == 2:
        _HOST.flappie_matrix_sync(pm)
    return np.ctypeslib.as_array(m.f, shape=(m.nc, m.stride))[:, : m.nr].copy()


# ------------------------------------------------------------------------------------ CPU
def test_registry_names(host):
    # networks.c:21-83 and networks.h:18-29
    names = ["r941_native", "r941_rna002", "r941_5mC", "r103_native"]
    for i, n in enumerate(names):
        assert host.get_flappie_model_type(n.encode()) == i
        assert host.flappie_model_string(i).decode() == n
        assert host.flappie_model_description(i).decode() == M.REGISTRY[n]["description"]
    assert host.get_flappie_model_type(b"rle_r941_native") == 5
    assert host.get_flappie_model_type(b"r10C_pcr") == 4          # FLAPPIE_MODEL_INVALID


def test_matrix_type(host):
    # flappie_matrix.c:20-61,142-148
    m = host.make_flappie_matrix(5, 9)
    assert (m.contents.nr, m.contents.nrq, m.contents.nc, m.contents.stride) == (5, 2, 9, 8)
    assert C.addressof(m.contents.f.contents) % 16 == 0
    assert not np.ctypeslib.as_array(m.contents.f, shape=(9, 8)).any()
    same = host.remake_flappie_matrix(m, 5, 9)
    assert C.addressof(same.contents) == C.addressof(m.contents)      # same shape: reused
    other = host.remake_flappie_matrix(same, 6, 9)
    assert other.contents.nr == 6
    assert not host.free_flappie_matrix(other)                        # returns NULL
    x = np.arange(10, dtype=np.float32)
    mm = host.mat_from_array(_f(x), 5, 2)
    assert np.array_equal(_dense(mm), x.reshape(2, 5))
    host.free_flappie_matrix(mm)


@pytest.mark.gpu
def test_signal_prep_matches_reference_fixtures_and_oracle(host):
    """The reference's own fixtures for this path (test_flappie_signal.c:67-111), through the C names, on the GPU."""
    from oracle import ffo
    sig = np.load(os.path.join(HERE, "golden", "signal_fixtures.npz"))
    unit = np.float32(1373.41) / np.float32(8192.0)
    raw = ((sig["raw"].astype(np.float32) + np.float32(16.0)) * unit).astype(np.float32)
    # trim_raw_by_mad + fixed trims == trimmed_signal.crp (test_flappie_signal.c:67-96)
    buf = raw.copy()
    out = host.trim_raw_by_mad(RawTable(None, buf.size, 0, buf.size, _f(buf)), 100, 0.0)
    assert (out.start, out.end) == (0, (buf.size // 100) * 100)
    assert np.abs(buf[out.start + 200: out.end - 10] - sig["trimmed"]).max() <= 1e-4
    x = sig["trimmed"].copy()
    host.medmad_normalise_array(_f(x), x.size)
    assert np.abs(x - sig["normalised"]).max() <= 1e-5
    # bit-identical to the oracle (itself checked against the compiled reference in test_oracle_cpu.py)
    y = sig["trimmed"].copy()
    ffo.lib().fo_medmad_normalise_array(_f(y), y.size)
    assert np.array_equal(x, y)
    a = np.array([0, 1, 2, 3, 4], dtype=np.float32)
    assert host.medianf(_f(a), 5) == 2.0 and abs(host.medianf(_f(a), 4) - 1.5) <= 1e-5   # test_util.c:32-42


@pytest.mark.parametrize("name", ["r941_native", "r941_5mC"])
def test_c_mdl_reader(host, tmp_path, name):
    reg = M.REGISTRY[name]
    mdl = M.synthetic_model(reg["kind"], 32, seed=5, ident=reg["ident"])
    path = str(tmp_path / "m.mdl")
    M.write_mdl(path, mdl)
    h = host.mdl_load(path.encode())
    assert h
    names = M.tensor_names(reg["kind"], reg["ident"])
    pairs = [(names["FF"] + "W", mdl.FF_W), (names["rnn2"] + "sW", mdl.rnns[2].sW), (names["conv1"] + "W", mdl.convs[0].W),
             (names["rnn4"] + "b", mdl.rnns[4].b)]
    for nm, mat in pairs:
        pm = host.mdl_matrix(h, nm.encode())
        assert pm, nm
        assert (pm.contents.nr, pm.contents.nc, pm.contents.stride) == (mat.nr, mat.nc, mat.stride)
        assert np.array_equal(np.ctypeslib.as_array(pm.contents.f, shape=(mat.nc, mat.stride)), mat.data)
    assert host.mdl_define(h, (names["conv1"] + "stride").encode(), -1) == mdl.convs[0].stride
    assert not host.mdl_matrix(h, b"no_such_tensor")
    host.mdl_free(h)
    # a git-LFS pointer stub (what the reference checkout holds) is rejected, not parsed
    stub = tmp_path / "stub.mdl"
    stub.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    assert not host.mdl_load(str(stub).encode())


# ------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("models")
    mdls = {}
    for name, header, H in (("r941_native", "flipflop5_r941native.h", 64), ("r941_5mC", "flipflop_r941native5mC.h", 48)):
        reg = M.REGISTRY[name]
        mdl = M.synthetic_model(reg["kind"], H, seed=7, ident=reg["ident"])
        M.write_mdl(str(d / header), mdl)
        mdls[name] = mdl
    return str(d), mdls


@pytest.mark.gpu
def test_calculate_post_sequence_matches_oracle(host, model_dir):
    """The exact call sequence of flappie.c:245-316 through the reference-named C functions."""
    from oracle import ffo
    d, mdls = model_dir
    os.environ["FLAPPIE_MODEL_DIR"] = d
    try:
        for enum, name, T in ((0, "r941_native", 2000), (2, "r941_5mC", 1500)):
            om = ffo.OracleModel(mdls[name])
            raw = np.random.default_rng(3 + enum).standard_normal(T + 100).astype(np.float32)
            rt = RawTable(None, raw.size, 60, 60 + T, _f(raw))
            ref = om.basecall(raw[60:60 + T])
            trans = host.calculate_transitions(rt, 1.0, enum)
            assert trans, "calculate_transitions returned NULL"
            nblock = trans.contents.nc
            assert np.abs(_dense(trans) - ref["trans"]).max() <= 1e-4
            post = host.transpost_crf_flipflop(trans, True)
            assert np.abs(_dense(post) - ref["post"]).max() <= 2e-4
            path = np.zeros(nblock + 2, dtype=np.int32)
            qpath = np.zeros(nblock + 2, dtype=np.float32)
            score = host.decode_crf_flipflop(post, False, path.ctypes.data_as(C.POINTER(C.c_int)), _f(qpath))
            assert np.array_equal(path[: nblock + 1], ref["path"])
            assert abs(score - ref["score"]) <= 1e-2
            idx = np.zeros(nblock + 2, dtype=np.int32)
            n = host.change_positions(path.ctypes.data_as(C.POINTER(C.c_int)), nblock, idx.ctypes.data_as(C.POINTER(C.c_int)))
            nbase = mdls[name].nbase
            bases = "".join("ACGTZ"[path[i] % nbase] for i in idx[:n])
            assert bases == ref["basecall"]
            host.exp_activation_inplace(post)
            tr = host.trace_from_posterior(post)
            t = np.ctypeslib.as_array(tr.contents.f, shape=(nblock + 1, tr.contents.stride))[:, : tr.contents.nr]
            assert np.abs(t - ref["trace"]).max() <= 1
            host.free_flappie_imatrix(tr)
            host.free_flappie_matrix(post)
            host.free_flappie_matrix(trans)
        # NULL conventions (networks.c:540-541)
        empty = RawTable(None, 0, 0, 0, None)
        assert not host.calculate_transitions(empty, 1.0, 0)
        assert not host.calculate_transitions(RawTable(None, 10, 0, 10, _f(np.zeros(10, np.float32))), 1.0, 1)   # model file absent
    finally:
        host.flappie_hip_shutdown()
        del os.environ["FLAPPIE_MODEL_DIR"]


def _copy_counts(host, reset=False):
    c = (C.c_ulonglong * 5)()
    host.ffhip_copy_counts(c, 1 if reset else 0)
    return dict(h2d_calls=c[0], h2d_bytes=c[1], d2h_calls=c[2], d2h_bytes=c[3], d2h_largest=c[4])


@pytest.mark.gpu
def test_matrices_of_the_hot_path_stay_on_the_device(host, model_dir):
    """north_star: "flappie_matrix is backed by HIP device buffers" (type flappie_matrix.h:18-24).  calculate_transitions returns its
    scores as a device image; transpost_crf_flipflop, decode_crf_flipflop, exp_activation_inplace and trace_from_posterior
    (flappie.c:266-300) consume and produce device images -- between the signal going up and path / trace coming down NO matrix
    crosses PCIe (counted: the library's own copy accounting) -- and the results equal those of the host-image mode
    (FLAPPIE_HOST_MATRICES=1 / ffhip_set_matrix_policy(0)), which moves every matrix both ways as round 2 did."""
    d, mdls = model_dir
    os.environ["FLAPPIE_MODEL_DIR"] = d
    try:
        T = 3000
        raw = np.random.default_rng(11).standard_normal(T).astype(np.float32)
        rt = RawTable(None, raw.size, 0, T, _f(raw))
        got = {}
        for policy in (1, 0):
            host.ffhip_set_matrix_policy(policy)
            trans = host.calculate_transitions(rt, 1.0, 0)
            assert trans
            nblock, nparam, stride = trans.contents.nc, trans.contents.nr, trans.contents.stride
            matrix_bytes = nblock * stride * 4
            assert (trans.contents.dev_state, bool(trans.contents.dev)) == ((2, True) if policy else (0, False))
            _copy_counts(host, reset=True)
            post = host.transpost_crf_flipflop(trans, True)
            path = np.zeros(nblock + 2, dtype=np.int32)
            qpath = np.zeros(nblock + 2, dtype=np.float32)
            score = host.decode_crf_flipflop(post, False, path.ctypes.data_as(C.POINTER(C.c_int)), _f(qpath))
            host.exp_activation_inplace(post)
            tr = host.trace_from_posterior(post)
            c = _copy_counts(host)
            trace = np.ctypeslib.as_array(tr.contents.f, shape=(nblock + 1, tr.contents.stride))[:, : tr.contents.nr].copy()
            if policy:
                assert post.contents.dev_state == 2
                # down: path, qpath, score, trace -- and nothing as large as a matrix; up: nothing at all
                assert c["h2d_calls"] == 0 and c["d2h_calls"] == 4, c
                assert c["d2h_largest"] == (nblock + 1) * tr.contents.nr * 4 < matrix_bytes and c["d2h_bytes"] < matrix_bytes, (c, matrix_bytes)
            else:
                assert post.contents.dev_state == 0 and not post.contents.dev
                assert c["h2d_bytes"] >= 4 * matrix_bytes and c["d2h_bytes"] >= 2 * matrix_bytes       # round 2's traffic: every call moves its matrix up, two move one down
            got[policy] = (_dense(trans), _dense(post), path.copy(), qpath.copy(), score, trace)
            assert post.contents.dev_state == (1 if policy else 0)                                  # _dense synchronised it: host and device equal
            host.free_flappie_imatrix(tr)
            host.free_flappie_matrix(post)
            host.free_flappie_matrix(trans)
        for a, b in zip(got[1], got[0]):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
        # operators of layers.h / flappie_matrix.h follow the same rule: on a matrix that was put on the device they work in place there,
        # the library's own readers (array_from_flappie_matrix) see the current values, and a host-side write is declared
        host.ffhip_set_matrix_policy(1)
        x = np.random.default_rng(5).standard_normal((7, 10)).astype(np.float32)          # 7 columns of 10 rows (stride 12)
        m = host.mat_from_array(_f(x), 10, 7)
        assert host.flappie_matrix_to_device(m) and m.contents.dev_state == 1
        _copy_counts(host, reset=True)
        host.tanh_activation_inplace(m)
        assert m.contents.dev_state == 2 and _copy_counts(host)["d2h_calls"] == 0
        assert np.array_equal(np.ctypeslib.as_array(m.contents.f, shape=(7, 12))[:, :10], x)          # the host image is the OLD one until synchronised
        arr = host.array_from_flappie_matrix(m)                                                       # synchronises
        now = np.ctypeslib.as_array(arr, shape=(7, 10)).copy()
        C.CDLL(None).free(arr)
        ref = host.mat_from_array(_f(x), 10, 7)
        host.tanh_activation_inplace(ref)                                                             # host-image path: upload, kernel, download
        assert ref.contents.dev_state == 0 and np.array_equal(now, _dense(ref)) and not np.array_equal(now, x)
        np.ctypeslib.as_array(m.contents.f, shape=(7, 12))[:] = 0.25
        host.flappie_matrix_host_changed(m)
        host.tanh_activation_inplace(m)
        assert m.contents.dev_state == 0 and np.allclose(_dense(m), np.tanh(0.25), atol=1e-6)
        host.free_flappie_matrix(m)
        host.free_flappie_matrix(ref)
    finally:
        host.ffhip_set_matrix_policy(1)
        host.flappie_hip_shutdown()
        del os.environ["FLAPPIE_MODEL_DIR"]


@pytest.mark.gpu
def test_transition_matrix_released_with_a_plain_free(host, model_dir):
    """flappie.c:281 releases the transition matrix with free(), not free_flappie_matrix(): the device image of such a struct is found again
    through its address (ffhip_dev_remember / ffhip_dev_forget, flappie_matrix.c) when malloc hands that address to the next matrix -- which a C
    program's allocator does within a read or two (the struct's size class is its own), and which this process, whose interpreter allocates
    between the calls, does only now and then: what is held here is what must hold EITHER way.  The reference's calculate_post sequence
    40 times: no buffer may ever sit twice in a free list; an image released by another path first (a host-side write, an operator replacing
    a stale output image) leaves no record behind and is not released again when its address comes back; every buffer the pool has
    allocated is in a free list, remembered for an address, or owned by a live matrix (none at the end)."""
    d, mdls = model_dir
    os.environ["FLAPPIE_MODEL_DIR"] = d
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    host.ffhip_debug_pool_state.argtypes = [C.POINTER(C.c_ulonglong)]
    host.flappie_matrix_host_changed.argtypes = [C.POINTER(CMat)]

    def state():
        out = (C.c_ulonglong * 4)()
        host.ffhip_debug_pool_state(out)
        return dict(zip(("buffers", "free", "owners", "twice"), (int(v) for v in out)))

    try:
        host.ffhip_set_matrix_policy(1)
        rng = np.random.default_rng(17)
        seen = []
        for it in range(40):
            T = int(rng.integers(1500, 3000))
            raw = rng.standard_normal(T).astype(np.float32)
            rt = RawTable(None, raw.size, 0, T, _f(raw))
            trans = host.calculate_transitions(rt, 1.0, 0)
            assert trans and trans.contents.dev_state == 2
            post = host.transpost_crf_flipflop(trans, True)
            if it % 3 == 1:
                host.flappie_matrix_host_changed(trans)          # the image goes back to the pool here; the record must go with it
                assert not trans.contents.dev
            elif it % 3 == 2:
                host.tanh_activation_inplace(trans)              # (works on the image in place: still owned)
            libc.free(C.cast(trans, C.c_void_p))                 # flappie.c:281 -- struct gone, data.f leaks as in the reference
            path = np.zeros(T, dtype=np.int32)
            qpath = np.zeros(T, dtype=np.float32)
            host.decode_crf_flipflop(post, False, path.ctypes.data_as(C.POINTER(C.c_int)), _f(qpath))
            host.free_flappie_matrix(post)
            s = state()
            assert s["twice"] == 0, (it, s)
            seen.append(s)
        print("pool after 40 reads:", seen[-1], "after 10:", seen[9])
        # 40 reads: 13 images were released by the host-side write (no record), at most 27 can still be remembered; nothing else is outstanding
        assert seen[-1]["owners"] <= 27 and seen[-1]["buffers"] == seen[-1]["free"] + seen[-1]["owners"], seen[-1]
    finally:
        host.flappie_hip_shutdown()
        del os.environ["FLAPPIE_MODEL_DIR"]


@pytest.mark.gpu
def test_flappie_lite_fastq(model_dir, tmp_path):
    """C driver end to end: float32 signal files -> FASTQ; compares calls with the oracle on the same
    prepared signal and checks the header fields of flappie_output.c:112-116."""
    from oracle import ffo
    d, mdls = model_dir
    rng = np.random.default_rng(12)
    files = []
    for i, n in enumerate((3000, 3000, 2300)):
        x = (rng.standard_normal(n) * 12 + 90).astype(np.float32)
        p = tmp_path / ("read%d.f32" % i)
        x.tofile(p)
        files.append((str(p), x))
    env = dict(os.environ, FLAPPIE_MODEL_DIR=d)
    out = subprocess.run([os.path.join(ROOT, "flappie_amd", "flappie_lite"), "--model", "r941_native"] + [f for f, _ in files],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    assert len(lines) == 4 * len(files)
    om = ffo.OracleModel(mdls["r941_native"])
    got = {}
    for k in range(0, len(lines), 4):
        assert lines[k].startswith("@") and lines[k + 2] == "+"
        fn = lines[k][1:].split("  {")[0]
        got[fn] = (lines[k], lines[k + 1], lines[k + 3])
    for f, x in files:
        s, e = C.c_size_t(0), C.c_size_t(x.size)
        assert ffo.lib().fo_trim_and_segment_raw(_f(x), x.size, C.byref(s), C.byref(e), 200, 10, 100, 0.0) == 0
        y = x[s.value:e.value].copy()
        ffo.lib().fo_medmad_normalise_array(_f(y), y.size)
        ref = om.basecall(y)
        hdr, bases, quals = got[f]
        assert bases == ref["basecall"] and quals == ref["quality"]
        assert '"nblock" : %d' % ref["nblock"] in hdr and '"trim" : [ %d, %d ]' % (s.value, e.value) in hdr
