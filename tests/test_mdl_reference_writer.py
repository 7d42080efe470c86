"""Row A1 held to the REFERENCE's own writer.  tests/golden/ref_writer_*.mdl are the bytes that
misc/taiyaki_flipflop5_guppy.py / taiyaki_rle5.py / taiyaki_flipflop_guppy.py (their cformatM, cformatV, print_lstm, print_gru, print_convolution,
imported where they lie by tests/golden/make_mdl_fixture.py, and each script's own __main__) print for flappie_amd.model.synthetic_model(kind, H, seed 11).
Held here:
  * the run-time C parser (flappie_amd/host/mdl_loader.c) reads such a file to exactly the arrays that went in, every tensor and every #define;
  * a C COMPILER reads it to the same arrays -- the files are C headers, that is how the reference consumes them (networks.c:10-14);
  * flappie_amd.model.write_mdl emits the same text byte for byte, and load_mdl reads it back;
  * (GPU) the engine, loading the files through the C parser under the reference's header names, calls reads as the oracle does."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from flappie_amd import model as M
from test_host_layer import RawTable, _dense, _f, host  # noqa: F401  (`host`: the ctypes view of libflappie_host.so, a fixture)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
HOSTLIB = os.path.join(ROOT, "flappie_amd", "libflappie_host.so")
SEED = 11
CASES = [("ref_writer_lstm5_h16.mdl", M.NET_LSTM5, "r941native", 16), ("ref_writer_grumod5_h16.mdl", M.NET_GRUMOD5, "r941native5mC", 16),
         ("ref_writer_rle5_h8.mdl", M.NET_LSTM5_RLE, "r941native", 8)]


class CMat(C.Structure):
    _fields_ = [("nr", C.c_size_t), ("nrq", C.c_size_t), ("nc", C.c_size_t), ("stride", C.c_size_t),
                ("f", C.POINTER(C.c_float)), ("dev", C.c_void_p), ("dev_state", C.c_int)]


def _tensors(mdl):
    """(symbol name, Mat) of every tensor the file holds, in file order"""
    names = M.tensor_names(mdl.kind, mdl.ident)
    out = []
    for i, c in enumerate(mdl.convs):
        out += [(names["conv%d" % (i + 1)] + "W", c.W), (names["conv%d" % (i + 1)] + "b", c.b)]
    for i, r in enumerate(mdl.rnns):
        out += [(names["rnn%d" % i] + "iW", r.iW), (names["rnn%d" % i] + "sW", r.sW), (names["rnn%d" % i] + "b", r.b)]
    return out + [(names["FF"] + "W", mdl.FF_W), (names["FF"] + "b", mdl.FF_b)]


@pytest.mark.parametrize("fname,kind,ident,hidden", CASES)
def test_write_mdl_emits_the_reference_writers_bytes(tmp_path, fname, kind, ident, hidden):
    mdl = M.synthetic_model(kind, hidden, seed=SEED, ident=ident)
    out = str(tmp_path / "own.mdl")
    M.write_mdl(out, mdl)
    own, ref = open(out, "rb").read(), open(os.path.join(GOLD, fname), "rb").read()
    assert len(ref) > 50000
    assert own == ref, "first difference at byte %d" % next(i for i in range(min(len(own), len(ref))) if own[i] != ref[i])


@pytest.mark.parametrize("fname,kind,ident,hidden", CASES)
def test_python_reader_on_the_reference_writers_file(fname, kind, ident, hidden):
    mdl = M.synthetic_model(kind, hidden, seed=SEED, ident=ident)
    back = M.load_mdl(os.path.join(GOLD, fname), kind, ident)
    assert [(c.stride, c.winlen, c.nf) for c in back.convs] == [(c.stride, c.winlen, c.nf) for c in mdl.convs]
    for (na, a), (nb, b) in zip(_tensors(mdl), _tensors(back)):
        assert na == nb and (a.nr, a.nc) == (b.nr, b.nc) and np.array_equal(a.data, b.data), na


@pytest.mark.parametrize("fname,kind,ident,hidden", CASES)
def test_c_parser_on_the_reference_writers_file(fname, kind, ident, hidden):
    if not os.path.exists(HOSTLIB):
        pytest.fail("libflappie_host.so not built: run __graft_entry__.build()")
    L = C.CDLL(HOSTLIB)
    L.mdl_load.restype = C.c_void_p
    L.mdl_load.argtypes = [C.c_char_p]
    L.mdl_matrix.restype = C.POINTER(CMat)
    L.mdl_matrix.argtypes = [C.c_void_p, C.c_char_p]
    L.mdl_define.restype = C.c_int
    L.mdl_define.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.mdl_free.argtypes = [C.c_void_p]
    mdl = M.synthetic_model(kind, hidden, seed=SEED, ident=ident)
    h = L.mdl_load(os.path.join(GOLD, fname).encode())
    assert h, "the C parser rejects the reference writer's file"
    for name, mat in _tensors(mdl):
        pm = L.mdl_matrix(h, name.encode())
        assert pm, name
        assert (pm.contents.nr, pm.contents.nrq, pm.contents.nc, pm.contents.stride) == (mat.nr, mat.nrq, mat.nc, mat.stride), name
        assert np.array_equal(np.ctypeslib.as_array(pm.contents.f, shape=(mat.nc, mat.stride)), mat.data), name
    names = M.tensor_names(kind, ident)
    for i, c in enumerate(mdl.convs):
        p = names["conv%d" % (i + 1)]
        assert L.mdl_define(h, (p + "stride").encode(), -1) == c.stride          # what networks.c:222-228, 298 take from the header
        if kind == M.NET_GRUMOD5:                                              # the GRU script's own names (taiyaki_flipflop_guppy.py:100-103)
            assert L.mdl_define(h, (ident + "_nfilter").encode(), -1) == c.W.nc
            assert L.mdl_define(h, ("_" + p + "winlen").encode(), -1) == c.winlen
        else:
            assert L.mdl_define(h, (p + "nfilter").encode(), -1) == c.W.nc
            assert L.mdl_define(h, (p + "winlen").encode(), -1) == c.winlen
    L.mdl_free(h)


@pytest.mark.parametrize("fname,kind,ident,hidden", CASES)
def test_a_c_compiler_reads_the_same_arrays(tmp_path, fname, kind, ident, hidden):
    """The reference consumes these files with #include (networks.c:10-14): compile one as the C header it is -- against include/flappie_matrix.h's _Mat, whose
    first five fields are the reference's -- and dump every tensor; the compiler's reading of the hex floats is the yardstick for the run-time parser."""
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    mdl = M.synthetic_model(kind, hidden, seed=SEED, ident=ident)
    inc = tmp_path / "inc" / "models"
    inc.mkdir(parents=True)
    shutil.copy(os.path.join(GOLD, fname), inc / "model.h")
    # the header asks for "../util.h" for the matrix type (util.h:13 includes flappie_matrix.h): hand it this repository's header under that name
    (tmp_path / "inc" / "util.h").write_text('#include "flappie_matrix.h"\n')
    tens = _tensors(mdl)
    body = "".join('  dump("%s", %s);\n' % (n, n) for n, _ in tens)
    (tmp_path / "dump.c").write_text(
        '#include <stdio.h>\n#include "models/model.h"\n'
        'static void dump(const char *name, const_flappie_matrix m) {\n'
        '  printf("%s %zu %zu %zu %zu\\n", name, m->nr, m->nrq, m->nc, m->stride);\n'
        '  fwrite(m->data.f, sizeof(float), m->nc * m->stride, stdout); printf("\\n");\n}\n'
        'int main(void) {\n' + body + '  return 0;\n}\n')
    exe = str(tmp_path / "dump")
    subprocess.run([gcc, "-std=gnu11", "-I", str(tmp_path / "inc"), "-I", os.path.join(ROOT, "include"), str(tmp_path / "dump.c"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, check=True).stdout
    pos = 0
    for name, mat in tens:
        eol = out.index(b"\n", pos)
        got_name, nr, nrq, nc, stride = out[pos:eol].decode().split()
        assert got_name == name and (int(nr), int(nrq), int(nc), int(stride)) == (mat.nr, mat.nrq, mat.nc, mat.stride), name
        n = int(nc) * int(stride) * 4
        arr = np.frombuffer(out[eol + 1: eol + 1 + n], dtype=np.float32).reshape(mat.nc, mat.stride)
        assert np.array_equal(arr, mat.data), name
        pos = eol + 1 + n + 1


@pytest.mark.gpu
def test_engine_calls_reads_from_the_reference_writers_files(host, tmp_path):
    """the files under the names networks.c:10-14 includes them by, through FLAPPIE_MODEL_DIR -> mdl_loader.c -> the engine: calls equal to the oracle's"""
    from oracle import ffo
    L = host
    for fname, header in (("ref_writer_lstm5_h16.mdl", "flipflop5_r941native.h"), ("ref_writer_grumod5_h16.mdl", "flipflop_r941native5mC.h")):
        shutil.copy(os.path.join(GOLD, fname), tmp_path / header)
    os.environ["FLAPPIE_MODEL_DIR"] = str(tmp_path)
    try:
        for enum, kind, ident, T in ((0, M.NET_LSTM5, "r941native", 1500), (2, M.NET_GRUMOD5, "r941native5mC", 1200)):
            om = ffo.OracleModel(M.synthetic_model(kind, 16, seed=SEED, ident=ident))
            raw = np.random.default_rng(40 + enum).standard_normal(T).astype(np.float32)
            ref = om.basecall(raw)
            rt = RawTable(None, raw.size, 0, T, _f(raw))
            trans = L.calculate_transitions(rt, 1.0, enum)
            assert trans, "calculate_transitions returned NULL"
            assert np.abs(_dense(trans) - ref["trans"]).max() <= 1e-4
            nblock = trans.contents.nc
            post = L.transpost_crf_flipflop(trans, True)
            path = np.zeros(nblock + 2, dtype=np.int32)
            qpath = np.zeros(nblock + 2, dtype=np.float32)
            L.decode_crf_flipflop(post, False, path.ctypes.data_as(C.POINTER(C.c_int)), _f(qpath))
            assert np.array_equal(path[: nblock + 1], ref["path"])
            L.free_flappie_matrix(post)
            L.free_flappie_matrix(trans)
    finally:
        L.flappie_hip_shutdown()
        del os.environ["FLAPPIE_MODEL_DIR"]
