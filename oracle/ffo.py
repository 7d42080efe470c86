"""ctypes binding of the CPU oracle (oracle/libff_oracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (flappie_amd/) must never import it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FoMat(C.Structure):
    _fields_ = [("nr", C.c_size_t), ("nrq", C.c_size_t), ("nc", C.c_size_t), ("stride", C.c_size_t),
                ("f", C.POINTER(C.c_float))]


class FoIMat(C.Structure):
    _fields_ = [("nr", C.c_size_t), ("nrq", C.c_size_t), ("nc", C.c_size_t), ("stride", C.c_size_t),
                ("f", C.POINTER(C.c_int32))]


class FoModel(C.Structure):
    _fields_ = [("kind", C.c_int), ("nconv", C.c_int),
                ("conv_W", C.POINTER(FoMat) * 3), ("conv_b", C.POINTER(FoMat) * 3),
                ("conv_stride", C.c_int * 3),
                ("rnn_iW", C.POINTER(FoMat) * 5), ("rnn_sW", C.POINTER(FoMat) * 5),
                ("rnn_b", C.POINTER(FoMat) * 5),
                ("FF_W", C.POINTER(FoMat)), ("FF_b", C.POINTER(FoMat))]


class FoReadResult(C.Structure):
    _fields_ = [("nblock", C.c_size_t), ("nbase", C.c_size_t), ("nstate", C.c_size_t),
                ("nparam", C.c_size_t), ("score", C.c_float), ("basecall_length", C.c_size_t)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libff_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ff_oracle.c", "ff_oracle.h", "cpu_ref.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs if os.path.exists(f)):
        subprocess.check_call(["make", "-C", _HERE, "libff_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    P = C.POINTER
    L.fo_set_dot_mode.restype = None
    L.fo_set_dot_mode.argtypes = [C.c_int]
    L.fo_make_mat.restype = P(FoMat)
    L.fo_make_mat.argtypes = [C.c_size_t, C.c_size_t]
    L.fo_free_mat.restype = P(FoMat)
    L.fo_free_mat.argtypes = [P(FoMat)]
    L.fo_free_imat.restype = P(FoIMat)
    L.fo_free_imat.argtypes = [P(FoIMat)]
    for name in ("fo_expf_cephes", "fo_logisticf", "fo_tanhf", "fo_eluf", "fo_logf_cephes"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float]
    L.fo_logsumexpf.restype = C.c_float
    L.fo_logsumexpf.argtypes = [C.c_float, C.c_float]
    L.fo_phredf.restype = C.c_char
    L.fo_phredf.argtypes = [C.c_float]
    for name in ("fo_swish_inplace", "fo_tanh_inplace", "fo_exp_inplace", "fo_row_normalise_inplace",
                 "fo_log_row_normalise_inplace", "fo_log_inplace", "fo_elu_inplace"):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [P(FoMat)]
    L.fo_robustlog_inplace.restype = None
    L.fo_robustlog_inplace.argtypes = [P(FoMat), C.c_float]
    L.fo_affine_map2.restype = P(FoMat)
    L.fo_affine_map2.argtypes = [P(FoMat)] * 5
    L.fo_convolution.restype = P(FoMat)
    L.fo_convolution.argtypes = [P(FoMat), P(FoMat), P(FoMat), C.c_size_t]
    L.fo_affine_map.restype = P(FoMat)
    L.fo_affine_map.argtypes = [P(FoMat), P(FoMat), P(FoMat)]
    L.fo_lstm.restype = P(FoMat)
    L.fo_lstm.argtypes = [P(FoMat), P(FoMat), C.c_int]
    L.fo_grumod.restype = P(FoMat)
    L.fo_grumod.argtypes = [P(FoMat), P(FoMat), C.c_int]
    L.fo_gru.restype = P(FoMat)
    L.fo_gru.argtypes = [P(FoMat), P(FoMat), P(FoMat), C.c_int, C.c_int, P(C.c_float)]
    L.fo_runlength_partition_function.restype = C.c_double
    L.fo_runlength_partition_function.argtypes = [P(FoMat)]
    L.fo_globalnorm_runlength.restype = P(FoMat)
    L.fo_globalnorm_runlength.argtypes = [P(FoMat), P(FoMat), P(FoMat), C.c_float]
    L.fo_partition_function.restype = C.c_double
    L.fo_partition_function.argtypes = [P(FoMat)]
    L.fo_globalnorm_flipflop.restype = P(FoMat)
    L.fo_globalnorm_flipflop.argtypes = [P(FoMat), P(FoMat), P(FoMat), C.c_float]
    L.fo_softplusf.restype = C.c_float
    L.fo_softplusf.argtypes = [C.c_float]
    L.fo_runlengthV2_partition_function.restype = C.c_double
    L.fo_runlengthV2_partition_function.argtypes = [P(FoMat)]
    L.fo_globalnorm_runlengthV2.restype = P(FoMat)
    L.fo_globalnorm_runlengthV2.argtypes = [P(FoMat), P(FoMat), P(FoMat), C.c_float]
    L.fo_decode_crf_runlength.restype = C.c_float
    L.fo_decode_crf_runlength.argtypes = [P(FoMat), P(C.c_int)]
    L.fo_transpost_crf_runlength.restype = P(FoMat)
    L.fo_transpost_crf_runlength.argtypes = [P(FoMat)]
    L.fo_dwmean.restype = C.c_float
    L.fo_dwmean.argtypes = [C.c_float, C.c_float, C.c_int]
    L.fo_runlengths_mean.restype = C.c_size_t
    L.fo_runlengths_mean.argtypes = [P(FoMat), P(C.c_int), P(C.c_int)]
    L.fo_runlengths_unit.restype = C.c_size_t
    L.fo_runlengths_unit.argtypes = [P(FoMat), P(C.c_int), P(C.c_int)]
    L.fo_runlength_to_basecall.restype = C.c_void_p          # (a malloc'ed string: freed through libc by the caller)
    L.fo_runlength_to_basecall.argtypes = [P(C.c_int), P(C.c_int), C.c_size_t]
    L.fo_decode_runlength.restype = C.c_float
    L.fo_decode_runlength.argtypes = [P(FoMat), P(C.c_int)]
    L.fo_posterior_runlength.restype = P(FoMat)
    L.fo_posterior_runlength.argtypes = [P(FoMat)]
    L.fo_runlength_records.restype = C.c_size_t
    L.fo_runlength_records.argtypes = [P(C.c_int), C.c_size_t, C.c_size_t, P(C.c_int), P(C.c_int), P(C.c_int)]
    L.fo_transitions.restype = P(FoMat)
    L.fo_transitions.argtypes = [P(C.c_float), C.c_size_t, C.c_size_t, C.c_float, P(FoModel)]
    L.fo_transpost.restype = P(FoMat)
    L.fo_transpost.argtypes = [P(FoMat), C.c_int]
    L.fo_decode_viterbi.restype = C.c_float
    L.fo_decode_viterbi.argtypes = [P(FoMat), C.c_int, P(C.c_int), P(C.c_float)]
    L.fo_change_positions.restype = C.c_size_t
    L.fo_change_positions.argtypes = [P(C.c_int), C.c_size_t, P(C.c_int)]
    L.fo_trace_from_posterior.restype = P(FoIMat)
    L.fo_trace_from_posterior.argtypes = [P(FoMat)]
    L.fo_basecall_read.restype = C.c_int
    L.fo_basecall_read.argtypes = [P(C.c_float), C.c_size_t, C.c_size_t, C.c_float, P(FoModel), C.c_int,
                                   P(FoReadResult), P(C.c_int), P(C.c_float), C.c_char_p, C.c_char_p,
                                   P(C.c_int32), P(C.c_float), P(C.c_float)]
    L.fo_quantilef.restype = None
    L.fo_quantilef.argtypes = [P(C.c_float), C.c_size_t, P(C.c_float), C.c_size_t]
    L.fo_medianf.restype = C.c_float
    L.fo_medianf.argtypes = [P(C.c_float), C.c_size_t]
    L.fo_madf.restype = C.c_float
    L.fo_madf.argtypes = [P(C.c_float), C.c_size_t, P(C.c_float)]
    L.fo_medmad_normalise_array.restype = None
    L.fo_medmad_normalise_array.argtypes = [P(C.c_float), C.c_size_t]
    L.fo_trim_raw_by_mad.restype = C.c_int
    L.fo_trim_raw_by_mad.argtypes = [P(C.c_float), P(C.c_size_t), P(C.c_size_t), C.c_size_t, C.c_float]
    L.fo_trim_and_segment_raw.restype = C.c_int
    L.fo_trim_and_segment_raw.argtypes = [P(C.c_float), C.c_size_t, P(C.c_size_t), P(C.c_size_t),
                                          C.c_size_t, C.c_size_t, C.c_size_t, C.c_float]
    _LIB = L
    return L


def find_openblas():
    """An LP64 OpenBLAS on this host, if any: the one scipy ships (`scipy_cblas_*` symbols), a distribution's libopenblas, conda's.
    (numpy's own `libscipy_openblas64_` is the ILP64 build: other symbol names, 64-bit integers -- not taken.)"""
    import glob
    import sys
    import sysconfig
    roots = {sysconfig.get_paths().get("purelib", ""), sysconfig.get_paths().get("platlib", "")} | {p for p in sys.path if p.endswith("-packages")}
    cands = []
    for r in sorted(x for x in roots if x):
        cands += sorted(glob.glob(os.path.join(r, "scipy.libs", "libscipy_openblas-*.so"))) + sorted(glob.glob(os.path.join(r, "scipy_openblas32", "lib", "*.so")))
    for pat in ("/usr/lib/x86_64-linux-gnu/libopenblas.so*", "/usr/lib/x86_64-linux-gnu/openblas-*/libopenblas.so*", "/usr/lib64/libopenblas.so*",
                "/opt/conda/lib/libopenblas.so*"):
        cands += sorted(glob.glob(pat))
    return cands


def use_dot_mode(mode: int):
    """Put the oracle library in dot mode `mode` (ff_oracle.c: 0 reference order, 1 double accumulators, 2 own vectorised kernels, 3 the
    GEMV / GEMM calls of the reference -- cblas_sgemv at layers.c:1009, cblas_sgemm at flappie_matrix.c:384 -- through a real OpenBLAS
    dlopen()ed from this host).  Mode 3 falls back to 2 when no library loads.  Returns (mode in force, (library file, its config) or None)."""
    L = lib()
    used = None
    if mode == 3:
        L.fo_blas_open.restype = C.c_int
        L.fo_blas_open.argtypes = [C.c_char_p]
        L.fo_blas_config.restype = C.c_char_p
        for path in find_openblas():
            if L.fo_blas_open(path.encode()) == 0:
                used = (os.path.basename(path), L.fo_blas_config().decode(errors="replace").strip())
                break
        if used is None:
            mode = 2
    L.fo_set_dot_mode(mode)
    return mode, used


class dot_mode:
    """with ffo.dot_mode(1): ...  -- how the oracle sums dot products inside the block (ff_oracle.c: 0 reference order in
    float = the oracle, 1 double accumulator = yardstick, 2 vectorised kernels = cpu_baseline timing)."""

    def __init__(self, mode: int):
        self.mode = mode

    def __enter__(self):
        lib().fo_set_dot_mode(self.mode)

    def __exit__(self, *exc):
        lib().fo_set_dot_mode(0)


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class HostMat:
    """Keeps a numpy [nc, stride] buffer alive behind an FoMat."""

    def __init__(self, nr: int, nc: int, data: np.ndarray):
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        stride = 4 * ((nr + 3) // 4)
        assert self.data.shape == (nc, stride), (self.data.shape, nc, stride)
        self.c = FoMat(nr, (nr + 3) // 4, nc, stride, _fptr(self.data))

    @staticmethod
    def from_model_mat(m) -> "HostMat":
        return HostMat(m.nr, m.nc, m.data)

    @staticmethod
    def from_dense(a: np.ndarray) -> "HostMat":
        """a[nc, nr] dense."""
        a = np.asarray(a, dtype=np.float32)
        nc, nr = a.shape
        stride = 4 * ((nr + 3) // 4)
        d = np.zeros((nc, stride), dtype=np.float32)
        d[:, :nr] = a
        return HostMat(nr, nc, d)

    @property
    def ptr(self):
        p = C.pointer(self.c)
        p._hostmat = self          # keep the numpy buffer alive for as long as the pointer object lives
        return p


def take(pmat, free: bool = True) -> np.ndarray:
    """Copy an oracle-owned fo_mat* into a dense [nc, nr] numpy array and free it."""
    if not pmat:
        raise RuntimeError("oracle returned NULL")
    m = pmat.contents
    a = np.ctypeslib.as_array(m.f, shape=(m.nc, m.stride)).copy()
    out = a[:, : m.nr].copy()
    if free:
        lib().fo_free_mat(pmat)
    return out


def take_i(pmat) -> np.ndarray:
    if not pmat:
        raise RuntimeError("oracle returned NULL")
    m = pmat.contents
    a = np.ctypeslib.as_array(m.f, shape=(m.nc, m.stride)).copy()
    out = a[:, : m.nr].copy()
    lib().fo_free_imat(pmat)
    return out


class OracleModel:
    """fo_model built from a flappie_amd.model.FlipflopModel (arrays are shared, not copied)."""

    def __init__(self, model):
        self.model = model
        self._keep = []
        m = FoModel()
        m.kind = model.kind
        m.nconv = len(model.convs)

        def mk(mat):
            h = HostMat.from_model_mat(mat)
            self._keep.append(h)
            return h.ptr

        for i, cv in enumerate(model.convs):
            m.conv_W[i] = mk(cv.W)
            m.conv_b[i] = mk(cv.b)
            m.conv_stride[i] = cv.stride
        for i, r in enumerate(model.rnns):
            m.rnn_iW[i] = mk(r.iW)
            m.rnn_sW[i] = mk(r.sW)
            m.rnn_b[i] = mk(r.b)
        m.FF_W = mk(model.FF_W)
        m.FF_b = mk(model.FF_b)
        self.c = m

    def transitions(self, signal: np.ndarray, temperature: float = 1.0) -> np.ndarray:
        sig = np.ascontiguousarray(signal, dtype=np.float32)
        p = lib().fo_transitions(_fptr(sig), 0, sig.size, temperature, C.byref(self.c))
        return take(p)          # [nblock, P]

    def runlength_call(self, signal: np.ndarray, temperature: float = 1.0, viterbi_only: bool = False):
        """runnie's calculate_post (runnie.c:241-316) for a prepared signal: parameters, posterior, path, score and the
        (base, shape, scale, dwell) records."""
        param = self.transitions(signal, temperature)                     # [nblock, P]
        nblock, P = param.shape
        nbase = self.model.nbase
        pm = HostMat.from_dense(param)
        scores = pm
        post = None
        if not viterbi_only:
            post = take(lib().fo_transpost_crf_runlength(pm.ptr))
            scores = HostMat.from_dense(post)
        path = np.zeros(nblock, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        score = lib().fo_decode_crf_runlength(scores.ptr, path.ctypes.data_as(ip))
        base, block, dwell = (np.zeros(nblock, dtype=np.int32) for _ in range(3))
        n = lib().fo_runlength_records(path.ctypes.data_as(ip), nblock, nbase, base.ctypes.data_as(ip), block.ctypes.data_as(ip),
                                       dwell.ctypes.data_as(ip))
        src = post if post is not None else param
        records = [("ACGTZ"[base[k]], float(src[block[k], base[k]]), float(src[block[k], nbase + base[k]]), int(dwell[k])) for k in range(n)]
        return dict(param=param, post=post, path=path, score=score, records=records)

    def basecall(self, signal: np.ndarray, temperature: float = 1.0, viterbi_only: bool = False,
                 want_trans: bool = True):
        sig = np.ascontiguousarray(signal, dtype=np.float32)
        nblock = self.model.nblock(sig.size)
        P, nstate = self.model.nparam, self.model.nstate
        path = np.zeros(nblock + 2, dtype=np.int32)
        qpath = np.zeros(nblock + 2, dtype=np.float32)
        bases = C.create_string_buffer(nblock + 2)
        quals = C.create_string_buffer(nblock + 2)
        trace = np.zeros((nblock + 1, nstate), dtype=np.int32)
        trans = np.zeros((nblock, P), dtype=np.float32) if want_trans else None
        post = np.zeros((nblock, P), dtype=np.float32) if want_trans else None
        res = FoReadResult()
        rc = lib().fo_basecall_read(_fptr(sig), 0, sig.size, temperature, C.byref(self.c),
                                    int(viterbi_only), C.byref(res),
                                    path.ctypes.data_as(C.POINTER(C.c_int)), _fptr(qpath), bases, quals,
                                    trace.ctypes.data_as(C.POINTER(C.c_int32)),
                                    _fptr(trans) if want_trans else None,
                                    _fptr(post) if want_trans else None)
        if rc != 0:
            raise RuntimeError("fo_basecall_read failed")
        return dict(nblock=nblock, score=float(res.score), path=path[: nblock + 1].copy(),
                    qpath=qpath[: nblock + 1].copy(), basecall=bases.value.decode(),
                    quality=quals.value.decode(), trace=trace, trans=trans, post=post)
