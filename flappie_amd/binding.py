"""ctypes binding of libffhip.so (the C-ABI of include/ffhip.h).

This is the Python mirror of the boundary used by tests/ and bench.py.  It contains no compute: if
the HIP library is missing or no gfx950 device is present it raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np

from . import model as M

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RUN_VITERBI_ONLY = 1
RUN_NO_TRACE = 2
RUN_NO_DECODE = 4
RUN_STEPWISE_RNN = 8
RUN_KEEP_ACTS = 16
RUN_UNFUSED_RNN = 32
RUN_F32_RNN = 64
RUN_FAST_GATES = 128
RUN_FAST_GATES2 = 256
RUN_EXACT_GATES = 512
NGROUP = 6
GROUP_NAMES = ("conv", "inproj", "recurrent", "head_crf", "posterior", "viterbi_assembly")


class FFHipError(RuntimeError):
    pass


class CMat(C.Structure):
    """`_Mat` of include/flappie_matrix.h"""
    _fields_ = [("nr", C.c_size_t), ("nrq", C.c_size_t), ("nc", C.c_size_t), ("stride", C.c_size_t),
                ("f", C.POINTER(C.c_float)), ("dev", C.c_void_p), ("dev_state", C.c_int)]


class CModelDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("nconv", C.c_int),
                ("conv_W", C.POINTER(CMat) * 3), ("conv_b", C.POINTER(CMat) * 3),
                ("conv_stride", C.c_int * 3),
                ("rnn_iW", C.POINTER(CMat) * 5), ("rnn_sW", C.POINTER(CMat) * 5),
                ("rnn_b", C.POINTER(CMat) * 5),
                ("FF_W", C.POINTER(CMat)), ("FF_b", C.POINTER(CMat))]


class CRawTable(C.Structure):
    """`raw_table` of include/flappie_structures.h"""
    _fields_ = [("uuid", C.c_char_p), ("n", C.c_size_t), ("start", C.c_size_t), ("end", C.c_size_t),
                ("raw", C.POINTER(C.c_float))]


def library_path() -> str:
    # FFHIP_BINDING_LIBRARY: another build of the same C-ABI (tests only: tools/test_hooks/libffhip_resweep.so, the library whose layer kernels re-sweep on purpose)
    return os.environ.get("FFHIP_BINDING_LIBRARY") or os.path.join(_HERE, "libffhip.so")


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise FFHipError("libffhip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.ffhip_last_error.restype = C.c_char_p
    L.ffhip_version.restype = C.c_char_p
    L.ffhip_device_count.restype = C.c_int
    L.ffhip_engine_create.restype = vp
    L.ffhip_engine_create.argtypes = [C.c_int]
    L.ffhip_engine_destroy.argtypes = [vp]
    L.ffhip_engine_synchronize.argtypes = [vp]
    L.ffhip_engine_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ffhip_engine_set_profiling.argtypes = [vp, C.c_int]
    L.ffhip_model_upload.restype = vp
    L.ffhip_model_upload.argtypes = [vp, C.POINTER(CModelDesc)]
    L.ffhip_model_free.argtypes = [vp]
    for fn in ("ffhip_model_hidden", "ffhip_model_nparam", "ffhip_model_nbase", "ffhip_model_launch_reads"):
        getattr(L, fn).restype = C.c_size_t
        getattr(L, fn).argtypes = [vp]
    L.ffhip_model_nblock.restype = C.c_size_t
    L.ffhip_model_nblock.argtypes = [vp, C.c_size_t]
    L.ffhip_batch_create.restype = vp
    L.ffhip_batch_create.argtypes = [vp, vp, C.c_int, C.c_size_t]
    L.ffhip_batch_destroy.argtypes = [vp]
    L.ffhip_batch_nblock.restype = C.c_size_t
    L.ffhip_batch_nblock.argtypes = [vp]
    L.ffhip_batch_set_reads.argtypes = [vp, C.POINTER(CRawTable)]
    L.ffhip_batch_set_signals.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t]
    L.ffhip_batch_run.argtypes = [vp, C.c_float, C.c_uint]
    L.ffhip_batch_run_pair.argtypes = [vp, vp, C.c_float, C.c_uint]
    L.ffhip_batch_paired.argtypes = [vp]
    L.ffhip_batch_read_nblock.restype = C.c_size_t
    L.ffhip_batch_read_nblock.argtypes = [vp, C.c_int]
    L.ffhip_batch_set_signals_ragged.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_size_t)]
    L.ffhip_prep_create.restype = vp
    L.ffhip_prep_create.argtypes = [vp, C.POINTER(CRawTable), C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_int, C.c_float]
    L.ffhip_prep_begin.restype = vp
    L.ffhip_prep_begin.argtypes = L.ffhip_prep_create.argtypes
    L.ffhip_prep_finish.argtypes = [vp]
    L.ffhip_prep_destroy.argtypes = [vp]
    L.ffhip_prep_range.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ffhip_prep_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ffhip_prep_get_signal.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.ffhip_batch_set_prepared.argtypes = [vp, vp, C.POINTER(C.c_int)]
    # packed batches (several reads to a row)
    L.ffhip_model_pack_gap.restype = C.c_size_t
    L.ffhip_model_pack_gap.argtypes = [vp]
    L.ffhip_model_packable.argtypes = [vp]
    L.ffhip_pack_plan.argtypes = [vp, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ffhip_batch_create_packed.restype = vp
    L.ffhip_batch_create_packed.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_int]
    L.ffhip_batch_set_prepared_packed.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ffhip_batch_set_signals_packed.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ffhip_batch_nreads.argtypes = [vp]
    L.ffhip_quantiles.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t]
    L.ffhip_medmad_normalise.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ffhip_mad.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ffhip_array_transform.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.c_float, C.c_float]
    L.ffhip_batch_finish.argtypes = [vp]
    L.ffhip_batch_basecall.restype = C.c_void_p
    L.ffhip_batch_basecall.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    L.ffhip_batch_quality.restype = C.c_void_p
    L.ffhip_batch_quality.argtypes = [vp, C.c_int]
    L.ffhip_batch_score.restype = C.c_float
    L.ffhip_batch_score.argtypes = [vp, C.c_int]
    L.ffhip_batch_get_path.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.ffhip_batch_get_transitions.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.ffhip_batch_get_posterior.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.ffhip_batch_get_trace.argtypes = [vp, C.c_int, C.POINTER(C.c_int32)]
    L.ffhip_batch_get_activation.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.ffhip_batch_profile.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.ffhip_batch_f32_reruns.argtypes = [vp]
    L.ffhip_engine_f32_reruns.restype = C.c_ulonglong
    L.ffhip_engine_f32_reruns.argtypes = [vp]
    _LIB = L
    return L


def _check(rc: int):
    if rc != 0:
        raise FFHipError("ffhip error %d: %s" % (rc, lib().ffhip_last_error().decode()))


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Engine:
    """One per GPU (ffhip_engine)."""

    def __init__(self, device: int = 0):
        self.h = lib().ffhip_engine_create(device)
        if not self.h:
            raise FFHipError(lib().ffhip_last_error().decode())
        self.device = device

    def info(self):
        name = C.create_string_buffer(64)
        ncu, clk = C.c_int(), C.c_int()
        _check(lib().ffhip_engine_info(self.h, name, 64, C.byref(ncu), C.byref(clk)))
        return dict(arch=name.value.decode(), ncu=ncu.value, clock_khz=clk.value)

    def synchronize(self):
        _check(lib().ffhip_engine_synchronize(self.h))

    def f32_reruns(self) -> int:
        return int(lib().ffhip_engine_f32_reruns(self.h))

    def set_profiling(self, on: bool):
        _check(lib().ffhip_engine_set_profiling(self.h, int(on)))

    def close(self):
        if self.h:
            lib().ffhip_engine_destroy(self.h)
            self.h = None


class DeviceModel:
    """Weights resident in HBM (ffhip_model), built from a flappie_amd.model.FlipflopModel."""

    def __init__(self, engine: Engine, mdl: M.FlipflopModel):
        self.engine = engine
        self.model = mdl
        keep = []

        def mk(mat: M.Mat):
            data = np.ascontiguousarray(mat.data, dtype=np.float32)
            cm = CMat(mat.nr, mat.nrq, mat.nc, mat.stride, _fptr(data), None, 0)
            keep.append((data, cm))
            return C.pointer(cm)

        d = CModelDesc()
        d.kind = mdl.kind
        d.nconv = len(mdl.convs)
        for i, cv in enumerate(mdl.convs):
            d.conv_W[i] = mk(cv.W)
            d.conv_b[i] = mk(cv.b)
            d.conv_stride[i] = cv.stride
        for i, r in enumerate(mdl.rnns):
            d.rnn_iW[i] = mk(r.iW)
            d.rnn_sW[i] = mk(r.sW)
            d.rnn_b[i] = mk(r.b)
        d.FF_W = mk(mdl.FF_W)
        d.FF_b = mk(mdl.FF_b)
        self.h = lib().ffhip_model_upload(engine.h, C.byref(d))
        del keep
        if not self.h:
            raise FFHipError(lib().ffhip_last_error().decode())

    @property
    def launch_reads(self) -> int:
        """reads per batch that keep every layer launch of this model full on this device (ffhip_model_launch_reads)"""
        return int(lib().ffhip_model_launch_reads(self.h))

    def close(self):
        if self.h:
            lib().ffhip_model_free(self.h)
            self.h = None


PREP_MEDMAD, PREP_DELTA, PREP_NONE = 0, 1, 2


class Prepared:
    """Raw reads trimmed and normalised on the device (ffhip_prep): trim_and_segment_raw + medmad_normalise_array."""

    def __init__(self, engine: "Engine", raws: List[np.ndarray], trim_start: int = 200, trim_end: int = 10,
                 varseg_chunk: int = 100, varseg_thresh: float = 0.0, mode: int = PREP_MEDMAD, delta: float = 0.0, begin_only: bool = False):
        """begin_only: ffhip_prep_begin (the work is enqueued, the call returns); `finish()` then waits -- ranges, statistics and signals are there after it"""
        self.engine = engine
        self.n = len(raws)
        arr = (CRawTable * self.n)()
        keep = [np.ascontiguousarray(r, dtype=np.float32) for r in raws]
        for i, r in enumerate(keep):
            arr[i] = CRawTable(None, r.size, 0, r.size, _fptr(r))
        fn = lib().ffhip_prep_begin if begin_only else lib().ffhip_prep_create
        self.h = fn(engine.h, arr, self.n, trim_start, trim_end, varseg_chunk, varseg_thresh, mode, delta)
        if not self.h:
            raise FFHipError(lib().ffhip_last_error().decode())

    def finish(self):
        _check(lib().ffhip_prep_finish(self.h))

    def range(self, i: int):
        s, e = C.c_size_t(0), C.c_size_t(0)
        _check(lib().ffhip_prep_range(self.h, i, C.byref(s), C.byref(e)))
        return s.value, e.value

    def stats(self, i: int):
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().ffhip_prep_stats(self.h, i, C.byref(a), C.byref(b)))
        return a.value, b.value

    def signal(self, i: int) -> np.ndarray:
        s, e = self.range(i)
        out = np.zeros(max(e - s, 0), dtype=np.float32)
        _check(lib().ffhip_prep_get_signal(self.h, i, _fptr(out)))
        return out

    def close(self):
        if self.h:
            lib().ffhip_prep_destroy(self.h)
            self.h = None


class Batch:
    """`nread` reads of `nsample` samples (ffhip_batch)."""

    def __init__(self, dmodel: DeviceModel, nread: int, nsample: int, max_reads: int = 0):
        """max_reads > 0: a PACKED batch -- nread rows of nsample samples that take up to max_reads reads, several to a row (set_signals_packed / set_prepared_packed)"""
        self.dmodel = dmodel
        self.nread, self.nsample = nread, nsample
        self.h = (lib().ffhip_batch_create_packed(dmodel.engine.h, dmodel.h, nread, nsample, max_reads) if max_reads > 0
                  else lib().ffhip_batch_create(dmodel.engine.h, dmodel.h, nread, nsample))
        if not self.h:
            raise FFHipError(lib().ffhip_last_error().decode())
        self.nblock = int(lib().ffhip_batch_nblock(self.h))
        self.P = dmodel.model.nparam
        self.nstate = dmodel.model.nstate

    def set_signals(self, signals: np.ndarray):
        s = np.ascontiguousarray(signals, dtype=np.float32)
        assert s.shape == (self.nread, self.nsample), s.shape
        _check(lib().ffhip_batch_set_signals(self.h, _fptr(s), s.shape[1]))

    def set_signals_ragged(self, signals: List[np.ndarray]):
        """reads of different lengths (each <= the batch's nsample); results are then per-read sized"""
        assert len(signals) == self.nread
        ld = max(int(x.size) for x in signals)
        buf = np.zeros((self.nread, ld), dtype=np.float32)
        lens = (C.c_size_t * self.nread)()
        for i, x in enumerate(signals):
            buf[i, :x.size] = x
            lens[i] = x.size
        _check(lib().ffhip_batch_set_signals_ragged(self.h, _fptr(buf), ld, lens))

    def read_nblock(self, read: int) -> int:
        return int(lib().ffhip_batch_read_nblock(self.h, read))

    def set_reads(self, raws: List[np.ndarray], starts: List[int]):
        """raw_table path: raws[i][starts[i]:starts[i]+nsample] is read i."""
        arr = (CRawTable * self.nread)()
        keep = []
        for i, (r, st) in enumerate(zip(raws, starts)):
            r = np.ascontiguousarray(r, dtype=np.float32)
            keep.append(r)
            arr[i] = CRawTable(None, r.size, st, st + self.nsample, _fptr(r))
        _check(lib().ffhip_batch_set_reads(self.h, arr))

    def set_prepared(self, prep: "Prepared", reads: List[int]):
        """device-to-device: prepared reads `reads` (all of this batch's length) become the batch's input"""
        assert len(reads) == self.nread
        idx = (C.c_int * self.nread)(*reads)
        _check(lib().ffhip_batch_set_prepared(self.h, prep.h, idx))

    def pack_plan(self, nsamples: List[int]):
        """places (ffhip_pack_plan: longest first, each into the emptiest row) of reads of these lengths in this batch's rows: (slot, block offset) per read, slot -1 where a read did not fit"""
        n = len(nsamples)
        ns = (C.c_size_t * n)(*[int(x) for x in nsamples])
        slot, off = (C.c_int * n)(), (C.c_int * n)()
        placed = lib().ffhip_pack_plan(self.dmodel.h, self.nread, self.nsample, n, ns, slot, off)
        if placed < 0:
            raise FFHipError(lib().ffhip_last_error().decode())
        return list(slot), list(off)

    def set_signals_packed(self, signals: List[np.ndarray], slots: List[int], offs: List[int]):
        """read i stands in row slots[i] from block offs[i] on; results are then indexed by read"""
        n = len(signals)
        keep = [np.ascontiguousarray(x, dtype=np.float32) for x in signals]
        ptrs = (C.POINTER(C.c_float) * n)(*[_fptr(x) for x in keep])
        ns = (C.c_size_t * n)(*[x.size for x in keep])
        _check(lib().ffhip_batch_set_signals_packed(self.h, n, ptrs, ns, (C.c_int * n)(*slots), (C.c_int * n)(*offs)))

    def set_prepared_packed(self, prep: "Prepared", reads: List[int], slots: List[int], offs: List[int]):
        n = len(reads)
        _check(lib().ffhip_batch_set_prepared_packed(self.h, prep.h, n, (C.c_int * n)(*reads), (C.c_int * n)(*slots), (C.c_int * n)(*offs)))

    def nreads(self) -> int:
        return int(lib().ffhip_batch_nreads(self.h))

    def run(self, temperature: float = 1.0, flags: int = 0):
        _check(lib().ffhip_batch_run(self.h, temperature, flags))

    def run_pair(self, other: "Batch", temperature: float = 1.0, flags: int = 0):
        """this batch and `other` (same model, same shape) with the recurrent layers of both as one launch per layer"""
        _check(lib().ffhip_batch_run_pair(self.h, other.h, temperature, flags))

    def paired(self) -> bool:
        return bool(lib().ffhip_batch_paired(self.h))

    def finish(self):
        _check(lib().ffhip_batch_finish(self.h))

    def basecall(self, read: int) -> str:
        n = C.c_size_t()
        p = lib().ffhip_batch_basecall(self.h, read, C.byref(n))
        if not p:
            raise FFHipError(lib().ffhip_last_error().decode())
        return C.string_at(p, n.value).decode()

    def quality(self, read: int) -> str:
        p = lib().ffhip_batch_quality(self.h, read)
        if not p:
            raise FFHipError(lib().ffhip_last_error().decode())
        return C.string_at(p).decode()

    def score(self, read: int) -> float:
        return float(lib().ffhip_batch_score(self.h, read))

    def path(self, read: int):
        path = np.zeros(self.read_nblock(read) + 1, dtype=np.int32)
        qpath = np.zeros(self.read_nblock(read) + 1, dtype=np.float32)
        _check(lib().ffhip_batch_get_path(self.h, read, path.ctypes.data_as(C.POINTER(C.c_int)), _fptr(qpath)))
        return path, qpath

    def transitions(self, read: int) -> np.ndarray:
        out = np.zeros((self.read_nblock(read), self.P), dtype=np.float32)
        _check(lib().ffhip_batch_get_transitions(self.h, read, _fptr(out)))
        return out

    def posterior(self, read: int) -> np.ndarray:
        out = np.zeros((self.read_nblock(read), self.P), dtype=np.float32)
        _check(lib().ffhip_batch_get_posterior(self.h, read, _fptr(out)))
        return out

    def trace(self, read: int) -> np.ndarray:
        out = np.zeros((self.read_nblock(read) + 1, self.nstate), dtype=np.int32)
        _check(lib().ffhip_batch_get_trace(self.h, read, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def activation(self, layer: int, read: int) -> np.ndarray:
        out = np.zeros((self.nblock, self.dmodel.model.hidden), dtype=np.float32)
        _check(lib().ffhip_batch_get_activation(self.h, layer, read, _fptr(out)))
        return out

    def rnn_path(self) -> int:
        """0 launch per step, 1 persistent recurrence + projection GEMM, 2 fused f32 layer kernel, 3 split-operand (fp16 x 2) layer kernel, 4 split-operand projection GEMM + recurrence-only split layer kernel"""
        L = lib()
        L.ffhip_batch_rnn_path.argtypes = [C.c_void_p]
        L.ffhip_batch_rnn_path.restype = C.c_int
        return int(L.ffhip_batch_rnn_path(self.h))

    def f32_reruns(self) -> int:
        """reads of the last run that left the split operand format's range and were run again on the f32 path (ffhip_batch_finish)"""
        return int(lib().ffhip_batch_f32_reruns(self.h))

    def profile(self):
        ms = (C.c_float * NGROUP)()
        ln = (C.c_int * NGROUP)()
        _check(lib().ffhip_batch_profile(self.h, ms, ln))
        return {GROUP_NAMES[i]: dict(ms=float(ms[i]), launches=int(ln[i])) for i in range(NGROUP)}

    def close(self):
        if self.h:
            lib().ffhip_batch_destroy(self.h)
            self.h = None


def basecall_reads(dmodel: DeviceModel, signals: np.ndarray, temperature: float = 1.0, flags: int = 0):
    """Convenience: one batch, returns list of (basecall, quality, score)."""
    b = Batch(dmodel, signals.shape[0], signals.shape[1])
    try:
        b.set_signals(signals)
        b.run(temperature, flags)
        b.finish()
        return [(b.basecall(i), b.quality(i), b.score(i)) for i in range(b.nread)]
    finally:
        b.close()
