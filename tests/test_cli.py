"""The `flappie` command line (host/flappie_cli.c) and its I/O helpers: FASTA/FASTQ/SAM record layout
(flappie_output.c:16-132), single-read fast5 input and --trace HDF5 output (fast5_interface.c:231-349),
option handling (flappie.c:42-235).  Needs libhdf5 to have been found at build time; skipped otherwise
(the HIP engine itself does not depend on HDF5)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from flappie_amd import model as M
from test_host_layer import CIMat, RawTable, ROOT, _f

FLAPPIE = os.path.join(ROOT, "flappie_amd", "flappie")
TOOL = os.path.join(ROOT, "flappie_amd", "fast5_tool")
FAST5LIB = os.path.join(ROOT, "flappie_amd", "libflappie_fast5.so")
HOSTLIB = os.path.join(ROOT, "flappie_amd", "libflappie_host.so")

needs_hdf5 = pytest.mark.skipif(not (os.path.exists(FLAPPIE) and os.path.exists(TOOL) and os.path.exists(FAST5LIB)),
                                reason="libhdf5 not found when the host layer was built")


class BasecallInfo(C.Structure):
    _fields_ = [("score", C.c_float), ("rt", RawTable), ("basecall", C.c_char_p), ("quality", C.c_char_p),
                ("basecall_length", C.c_size_t), ("trace", C.POINTER(CIMat)), ("pos", C.POINTER(C.c_int)),
                ("nblock", C.c_size_t)]


def write_fast5(path, read_id, raw_i16, digitisation=8192.0, offset=10.0, rng=1400.0, rate=4000.0):
    tmp = str(path) + ".i16"
    np.asarray(raw_i16, dtype="<i2").tofile(tmp)
    subprocess.run([TOOL, "write", str(path), read_id, repr(digitisation), repr(offset), repr(rng), repr(rate), tmp], check=True)
    os.unlink(tmp)


def dump_trace(path, group):
    out = subprocess.run([TOOL, "dump", str(path), group], capture_output=True, text=True, check=True).stdout.split("\n")
    n = int(out[0].split()[1])
    sig = np.array([float.fromhex(v) for v in out[1:1 + n]], dtype=np.float32)
    _, r, c = out[1 + n].split()
    tr = np.array([int(v) for v in out[2 + n:2 + n + int(r) * int(c)]], dtype=np.int32).reshape(int(r), int(c))
    return sig, tr


def synth_raw(rng, n):
    """int16 DAC values with a noisy head (so that trimming by MAD has something to find)"""
    x = rng.normal(500, 60, n)
    x[:300] = rng.normal(520, 4, 300)
    return np.clip(np.rint(x), 0, 8191).astype("<i2")


# ------------------------------------------------------------------------------------ CPU
def _cfile(libc, path, mode=b"w"):
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    return libc.fopen(str(path).encode(), mode)


def _cfloat(v):
    return "%f" % float(np.float32(v))


@pytest.mark.parametrize("uuid_primary", [True, False])
def test_record_layout(tmp_path, uuid_primary):
    L = C.CDLL(HOSTLIB)
    libc = C.CDLL(None)
    L.get_outformat.restype = C.c_int
    L.get_outformat.argtypes = [C.c_char_p]
    L.flappie_outformat_string.restype = C.c_char_p
    L.fprintf_format.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_bool, C.c_char_p, BasecallInfo]
    assert [L.get_outformat(s) for s in (b"fasta", b"fastq", b"sam", b"bam", b"FASTA")] == [0, 1, 2, 3, 3]
    assert L.get_outformat(None) == 3
    assert [L.flappie_outformat_string(i) for i in range(3)] == [b"fasta", b"fastq", b"sam"]
    res = BasecallInfo(score=np.float32(-123.5), basecall=b"ACGTTGCA", quality=b"!#%+5?IJ", basecall_length=8, nblock=37)
    res.rt = RawTable(uuid=b"u-1", n=4000, start=200, end=3990, raw=None)
    name = "u-1" if uuid_primary else "a.fast5"
    hdr = ('%sPRE_%s  { "filename" : "a.fast5", "uuid" : "u-1", "normalised_score" : %s,  "nblock" : 37,  "sequence_length" : 8,'
           '  "blocks_per_base" : %s, "nsample" : 4000, "trim" : [ 200, 3990 ] }\n')
    ns, bpb = _cfloat(np.float32(123.5) / np.float32(37)), _cfloat(np.float32(37) / np.float32(8))
    want = {0: hdr % (">", name, ns, bpb) + "ACGTTGCA\n",
            1: hdr % ("@", name, ns, bpb) + "ACGTTGCA\n+\n!#%+5?IJ\n",
            2: "PRE_%s\t4\t*\t0\t0\t*\t*\t0\t0\tACGTTGCA\t!#%%+5?IJ\nACGTTGCA\t!#%%+5?IJ\n" % name}
    for fmt in range(3):
        p = tmp_path / ("o%d" % fmt)
        fp = _cfile(libc, p)
        L.fprintf_format(fmt, fp, b"u-1", b"a.fast5", uuid_primary, b"PRE_", res)
        libc.fclose(fp)
        assert p.read_text() == want[fmt]
    # FASTQ without qualities: nothing written (flappie_output.c:107-110)
    res.quality = None
    p = tmp_path / "noq"
    fp = _cfile(libc, p)
    L.fprintf_format(1, fp, b"u-1", b"a.fast5", uuid_primary, b"", res)
    libc.fclose(fp)
    assert p.read_text() == ""


@pytest.fixture(scope="module")
def f5lib():
    L = C.CDLL(FAST5LIB)
    L.read_raw.restype = RawTable
    L.read_raw.argtypes = [C.c_char_p, C.c_bool]
    L.open_or_create_hdf5.restype = C.c_int64
    L.open_or_create_hdf5.argtypes = [C.c_char_p]
    L.write_summary.argtypes = [C.c_int64, C.c_char_p, BasecallInfo, C.c_uint64, C.c_int]
    L.H5Fclose.argtypes = [C.c_int64]                          # resolved through the library's libhdf5 dependency
    return L


@needs_hdf5
def test_read_raw(tmp_path, f5lib):
    rng = np.random.default_rng(3)
    raw = synth_raw(rng, 5000)
    p = tmp_path / "r.fast5"
    write_fast5(p, "0a1b2c3d-read", raw, 8192.0, 7.0, 1437.5)
    for scale in (False, True):
        rt = f5lib.read_raw(str(p).encode(), scale)
        assert rt.raw and rt.n == 5000 and (rt.start, rt.end) == (0, 5000) and rt.uuid == b"0a1b2c3d-read"
        got = np.ctypeslib.as_array(rt.raw, shape=(5000,)).copy()
        want = raw.astype(np.float32)
        if scale:
            want = (want + np.float32(7.0)) * (np.float32(1437.5) / np.float32(8192.0))
        np.testing.assert_array_equal(got, want)
    # failures: raw == NULL, no exception (fast5_interface.c:236-247)
    assert not f5lib.read_raw(str(tmp_path / "missing.fast5").encode(), True).raw
    junk = tmp_path / "junk.fast5"
    junk.write_bytes(b"not hdf5 at all")
    assert not f5lib.read_raw(str(junk).encode(), True).raw


@needs_hdf5
@pytest.mark.parametrize("level", [0, 1])
def test_write_summary_round_trip(tmp_path, f5lib, level):
    host = C.CDLL(HOSTLIB)
    host.make_flappie_imatrix.restype = C.POINTER(CIMat)
    host.make_flappie_imatrix.argtypes = [C.c_size_t, C.c_size_t]
    assert f5lib.open_or_create_hdf5(None) < 0
    p = tmp_path / "trace.hdf5"
    rng = np.random.default_rng(9)
    for k, name in enumerate((b"read_a", b"read_b")):          # second pass re-opens the existing file
        h = f5lib.open_or_create_hdf5(str(p).encode())
        assert h >= 0
        sig = rng.standard_normal(1000).astype(np.float32)
        nblock, nstate = 40 + k, 8
        tr = host.make_flappie_imatrix(nstate, nblock + 1)
        vals = rng.integers(0, 256, (nblock + 1, nstate)).astype(np.int32)
        np.ctypeslib.as_array(tr.contents.f, shape=(nblock + 1, tr.contents.stride))[:, :nstate] = vals
        res = BasecallInfo(score=0.0, basecall=b"A", quality=b"!", basecall_length=1, nblock=nblock, trace=tr)
        res.rt = RawTable(uuid=name, n=1000, start=100, end=900, raw=_f(sig))
        f5lib.write_summary(h, name, res, 50, level)
        assert f5lib.H5Fclose(h) >= 0
        s, t = dump_trace(p, name.decode())
        np.testing.assert_array_equal(s, sig[100:900])
        np.testing.assert_array_equal(t, vals)
    s, t = dump_trace(p, "read_a")                              # the first group survived the second open
    assert s.size == 800 and t.shape == (41, 8)


@needs_hdf5
@pytest.mark.parametrize("level,chunk", [(0, 50), (1, 50), (6, 200), (1, 5000)])
def test_summary_pack_reads_back_like_write_summary(tmp_path, f5lib, level, chunk):
    """the two-step writer (filters in summary_pack_create, outside libhdf5 -- what the binary's worker threads run -- and
    H5Dwrite_chunk in summary_pack_write) against write_summary: same groups, same signal, same trace after reading the files
    back; chunk sizes that do not divide the data, a chunk larger than the data, out-of-range trace values (the int32 -> u8
    conversion saturates)"""
    host = C.CDLL(HOSTLIB)
    host.make_flappie_imatrix.restype = C.POINTER(CIMat)
    host.make_flappie_imatrix.argtypes = [C.c_size_t, C.c_size_t]
    f5lib.summary_pack_create.restype = C.c_void_p
    f5lib.summary_pack_create.argtypes = [BasecallInfo, C.c_uint64, C.c_int]
    f5lib.summary_pack_write.argtypes = [C.c_int64, C.c_char_p, C.c_void_p]
    f5lib.summary_pack_free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(level * 100 + chunk)
    sig = rng.standard_normal(3000).astype(np.float32)
    nblock, nstate = 533, 10
    tr = host.make_flappie_imatrix(nstate, nblock + 1)
    vals = rng.integers(0, 256, (nblock + 1, nstate)).astype(np.int32)
    vals[3, 2], vals[7, 1] = 300, -5
    np.ctypeslib.as_array(tr.contents.f, shape=(nblock + 1, tr.contents.stride))[:, :nstate] = vals
    res = BasecallInfo(score=0.0, basecall=b"A", quality=b"!", basecall_length=1, nblock=nblock, trace=tr)
    res.rt = RawTable(uuid=b"r", n=3000, start=120, end=2950, raw=_f(sig))
    pa, pb = tmp_path / "a.hdf5", tmp_path / "b.hdf5"
    h = f5lib.open_or_create_hdf5(str(pa).encode())
    f5lib.write_summary(h, b"read", res, chunk, level)
    assert f5lib.H5Fclose(h) >= 0
    pack = f5lib.summary_pack_create(res, chunk, level)
    assert pack
    h = f5lib.open_or_create_hdf5(str(pb).encode())
    f5lib.summary_pack_write(h, b"read", pack)
    f5lib.summary_pack_free(pack)
    assert f5lib.H5Fclose(h) >= 0
    sa, ta = dump_trace(pa, "read")
    sb, tb = dump_trace(pb, "read")
    np.testing.assert_array_equal(sa, sig[120:2950])
    np.testing.assert_array_equal(sb, sa)
    np.testing.assert_array_equal(tb, ta)
    np.testing.assert_array_equal(tb, np.clip(vals, 0, 255))
    if level > 0:
        assert os.path.getsize(pb) <= 1.02 * os.path.getsize(pa) + 4096      # the same filters: the same size within the headers


@needs_hdf5
def test_cli_options_without_gpu(tmp_path):
    run = lambda *a: subprocess.run([FLAPPIE] + list(a), capture_output=True, text=True, timeout=60)  # noqa: E731
    r = run("--help")
    assert r.returncode == 0
    for opt in ("--delta", "--format", "--limit", "--model", "--output", "--prefix", "--temperature", "--trim", "--trace",
                "--viterbi", "--segmentation", "--hdf5-compression", "--hdf5-chunk", "--uuid", "--no-uuid", "--licence", "--batch"):
        assert opt in r.stdout, opt
    r = run("--model", "help")
    assert r.returncode == 0
    for name in ("r941_native", "r941_5mC", "r941_rna002", "r103_native"):          # rle_r941_native belongs to runnie, not listed (flappie.c:184-190)
        assert name in r.stdout
    assert run("--licence").returncode == 0
    assert run("--version").returncode == 0
    assert run("--format", "bam", "x.fast5").returncode != 0
    assert run("--model", "nonsense", "x.fast5").returncode != 0
    assert run("--temperature", "-1", "x.fast5").returncode != 0
    assert run("--segmentation", "50", "x.fast5").returncode != 0
    assert run().returncode != 0                                  # no input files: usage


# ------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def cli_inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    mdl = M.synthetic_model(M.NET_LSTM5, 48, seed=21, ident="r941native")
    M.write_mdl(str(d / "flipflop5_r941native.h"), mdl)
    reads = d / "reads"
    reads.mkdir()
    rng = np.random.default_rng(77)
    raws = {}
    for i, n in enumerate((4000, 4000, 3100, 4013, 2600, 3977, 1250, 3100)):     # mixed lengths: ragged batches inside the CLI
        raw = synth_raw(rng, n)
        write_fast5(reads / ("read_%02d.fast5" % i), "uuid-%04d" % i, raw)
        raws["read_%02d.fast5" % i] = ("uuid-%04d" % i, raw)
    return d, mdl, reads, raws


def _oracle_calls(mdl, raws, viterbi=False, temperature=1.0, trim=(200, 10)):
    from oracle import ffo
    om = ffo.OracleModel(mdl)
    out = {}
    for fn, (uuid, raw) in raws.items():
        x = (raw.astype(np.float32) + np.float32(10.0)) * (np.float32(1400.0) / np.float32(8192.0))
        s, e = C.c_size_t(0), C.c_size_t(x.size)
        assert ffo.lib().fo_trim_and_segment_raw(_f(x), x.size, C.byref(s), C.byref(e), trim[0], trim[1], 100, 0.0) == 0
        y = x[s.value:e.value].copy()
        ffo.lib().fo_medmad_normalise_array(_f(y), y.size)
        ref = om.basecall(y, viterbi_only=viterbi, temperature=temperature)
        ref.update(start=s.value, end=e.value, uuid=uuid, signal=y)
        out[fn] = ref
    return out


def _parse_fastq(text):
    lines = text[:-1].split("\n") if text.endswith("\n") else text.split("\n")      # (a read may have an EMPTY sequence and quality line: no strip())
    recs = []
    for k in range(0, len(lines), 4):
        assert lines[k][0] == "@" and lines[k + 2] == "+"
        recs.append((lines[k][1:].split("  {")[0], lines[k], lines[k + 1], lines[k + 3]))
    return recs


@needs_hdf5
@pytest.mark.gpu
def test_cli_fastq_and_trace_match_oracle(cli_inputs, tmp_path):
    d, mdl, reads, raws = cli_inputs
    env = dict(os.environ, FLAPPIE_MODEL_DIR=str(d))
    trace = tmp_path / "trace.hdf5"
    r = subprocess.run([FLAPPIE, "--model", "r941_native", "--trace", str(trace), "--batch", "2", str(reads)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    recs = _parse_fastq(r.stdout)
    ref = _oracle_calls(mdl, raws)
    assert sorted(x[0] for x in recs) == sorted(v["uuid"] for v in ref.values())      # --uuid is the default
    by_uuid = {v["uuid"]: (fn, v) for fn, v in ref.items()}
    for name, hdr, bases, quals in recs:
        fn, v = by_uuid[name]
        assert bases == v["basecall"] and quals == v["quality"], fn
        assert '"filename" : "%s"' % fn in hdr and '"nblock" : %d' % v["nblock"] in hdr
        assert '"trim" : [ %d, %d ]' % (v["start"], v["end"]) in hdr and '"nsample" : %d' % raws[fn][1].size in hdr
        score = float(hdr.split('"normalised_score" : ')[1].split(",")[0])
        assert abs(score - (-v["score"] / v["nblock"])) < 2e-4
        sig, tr = dump_trace(trace, name)
        np.testing.assert_array_equal(sig, v["signal"])
        assert tr.shape == v["trace"].shape and np.abs(tr - v["trace"]).max() <= 1


@needs_hdf5
@pytest.mark.gpu
def test_cli_formats_limit_reverse_viterbi(cli_inputs, tmp_path):
    d, mdl, reads, raws = cli_inputs
    env = dict(os.environ, FLAPPIE_MODEL_DIR=str(d))
    files = [str(reads / fn) for fn in sorted(raws)]
    ref = _oracle_calls(mdl, raws)
    refv = _oracle_calls(mdl, raws, viterbi=True, temperature=0.7, trim=(150, 20))

    out = tmp_path / "calls.fa"
    r = subprocess.run([FLAPPIE, "-f", "fasta", "--no-uuid", "-p", "run1_", "-l", "3", "-o", str(out)] + files,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout == "", r.stderr
    lines = out.read_text().strip().split("\n")
    assert len(lines) == 2 * 3                                      # --limit 3
    for k, fn in enumerate(sorted(raws)[:3]):                       # explicit files: input order kept
        assert lines[2 * k].startswith(">run1_%s  {" % fn)
        assert lines[2 * k + 1] == ref[fn]["basecall"]

    r = subprocess.run([FLAPPIE, "-f", "sam", "--reverse", "--viterbi", "--temperature", "0.7", "--trim", "150:20"] + files[:2],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert len(lines) == 4
    for k, fn in enumerate(sorted(raws)[:2]):
        v = refv[fn]
        cols = lines[2 * k].split("\t")
        assert cols[0] == v["uuid"] and cols[1:9] == ["4", "*", "0", "0", "*", "*", "0", "0"]
        assert cols[9] == v["basecall"][::-1] and cols[10] == v["quality"][::-1]
        assert lines[2 * k + 1] == cols[9] + "\t" + cols[10]

    # unreadable input: warning, other reads still called, exit status 0 (flappie.c:372-374)
    bad = tmp_path / "bad.fast5"
    bad.write_bytes(b"junk")
    r = subprocess.run([FLAPPIE, str(bad), files[0]], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "No basecall returned" in r.stderr
    assert len(_parse_fastq(r.stdout)) == 1


RUNNIE = os.path.join(ROOT, "flappie_amd", "runnie")


@needs_hdf5
@pytest.mark.gpu
def test_runnie_records_match_oracle(tmp_path):
    """runnie (runnie.c:241-316) on generated fast5 files: `# uuid` then one `base<TAB>shape<TAB>scale<TAB>dwell` line per
    emitted base, against the oracle's run-length pipeline on the same prepared signal."""
    from oracle import ffo
    mdl = M.synthetic_model(M.NET_LSTM5_RLE, 48, seed=9, ident="r941native")
    M.write_mdl(str(tmp_path / "runlength5_r941native.h"), mdl)
    reads = tmp_path / "reads"
    reads.mkdir()
    rng = np.random.default_rng(31)
    raws = {}
    for i, n in enumerate((3000, 2400, 3000)):
        raw = synth_raw(rng, n)
        write_fast5(reads / ("r%d.fast5" % i), "rle-%02d" % i, raw)
        raws["r%d.fast5" % i] = ("rle-%02d" % i, raw)
    env = dict(os.environ, FLAPPIE_MODEL_DIR=str(tmp_path))
    files = [str(reads / fn) for fn in sorted(raws)]
    for extra, viterbi in (([], False), (["--viterbi"], True)):
        r = subprocess.run([RUNNIE] + extra + files, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        blocks = r.stdout.strip().split("# ")[1:]
        assert len(blocks) == len(files)
        om = ffo.OracleModel(mdl)
        for blk, fn in zip(blocks, sorted(raws)):
            lines = blk.strip().split("\n")
            uuid, raw = raws[fn]
            assert lines[0] == uuid
            x = (raw.astype(np.float32) + np.float32(10.0)) * (np.float32(1400.0) / np.float32(8192.0))
            s, e = C.c_size_t(0), C.c_size_t(x.size)
            assert ffo.lib().fo_trim_and_segment_raw(_f(x), x.size, C.byref(s), C.byref(e), 200, 10, 100, 0.0) == 0
            y = x[s.value:e.value].copy()
            ffo.lib().fo_medmad_normalise_array(_f(y), y.size)
            ref = om.runlength_call(y, viterbi_only=viterbi)
            got = [ln.split("\t") for ln in lines[1:]]
            assert [g[0] for g in got] == [rec[0] for rec in ref["records"]]
            assert [int(g[3]) for g in got] == [rec[3] for rec in ref["records"]]
            assert sum(int(g[3]) for g in got) <= ref["param"].shape[0]
            for g, rec in zip(got, ref["records"]):
                assert abs(float(g[1]) - rec[1]) <= 2e-4 and abs(float(g[2]) - rec[2]) <= 2e-4


@needs_hdf5
@pytest.mark.gpu
def test_reader_processes_and_multi_gpu_script(cli_inputs, tmp_path):
    """BASELINE.json configs[2] at the size a 1-GPU box allows: a reads/ directory through (a) the binary with its reader processes
    (--readers 3: files read by forked children, order kept by the parent), (b) the same in-process (--readers 0), (c)
    tools/flappie_multi_gpu.sh with one slice -- the three outputs are byte-identical, an unreadable file among the inputs
    costs one warning and no reordering, and explicit file arguments come out in argument order (README.md:81-83,
    flappie.c:334-385)."""
    d, mdl, reads, raws = cli_inputs
    env = dict(os.environ, FLAPPIE_MODEL_DIR=str(d))
    big = tmp_path / "reads"
    big.mkdir()
    rng = np.random.default_rng(5)
    names = []
    for i in range(37):                                             # more files than readers x pipe depth matters for; mixed lengths
        fn = "read_%03d.fast5" % i
        if i == 11:
            (big / fn).write_bytes(b"not an hdf5 file")
        else:
            write_fast5(big / fn, "uuid-%04d" % i, synth_raw(rng, int(rng.integers(1500, 4200))))
        names.append(fn)
    files = [str(big / fn) for fn in names]
    outs = {}
    for tag, extra in (("procs", ["--readers", "3"]), ("inproc", ["--readers", "0"]), ("one", ["--readers", "1"])):
        r = subprocess.run([FLAPPIE, "--batch", "8", "--no-uuid"] + extra + files, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert r.stderr.count("No basecall returned") == 1 and "read_011.fast5" in r.stderr
        outs[tag] = r.stdout
    recs = _parse_fastq(outs["procs"])
    assert [x[0] for x in recs] == [fn for fn in names if fn != "read_011.fast5"]       # argument order, the bad file skipped
    assert outs["procs"] == outs["inproc"] == outs["one"]
    # the per-GPU launcher of configs[2] with one slice: same records as one process over the directory (sorted file order)
    script = os.path.join(ROOT, "tools", "flappie_multi_gpu.sh")
    r = subprocess.run([script, "1", str(tmp_path / "shard"), "--batch", "8", "--no-uuid", str(big)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "shard.0.fq").read_text() == outs["procs"]
    # two slices on the one GPU of this box, run one after the other by the script (FLAPPIE_SERIAL=1): slice g holds files g, g+2, ...
    r = subprocess.run([script, "2", str(tmp_path / "two"), "--batch", "8", "--no-uuid", str(big)], env=dict(env, FLAPPIE_SERIAL="1", FLAPPIE_DEVICES="0,0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    by_name = {x[0]: x for x in recs}
    for g in (0, 1):
        got = _parse_fastq((tmp_path / ("two.%d.fq" % g)).read_text())
        want = [fn for fn in names[g::2] if fn != "read_011.fast5"]
        assert [x[0] for x in got] == want
        assert all(x == by_name[x[0]] for x in got)


@needs_hdf5
@pytest.mark.gpu
def test_fast5_layouts_through_the_binary(cli_inputs, tmp_path):
    """single-read files in the layouts a fast5 comes in (contiguous / chunked + deflate + shuffle Signal, fixed and variable-length read_id, old and
    latest file format, more read groups, float32 / integer channel attributes, dense attribute storage = a file the libhdf5-free reader refuses) through
    the binary: the default (host/fast5_raw.c first, libhdf5 for what it refuses) gives the bytes FLAPPIE_DEBUG=hdf5_read (libhdf5 for every file) gives,
    with reader processes and in-process (fast5_interface.c:231-318)."""
    d, mdl, reads, raws = cli_inputs
    env = dict(os.environ, FLAPPIE_MODEL_DIR=str(d))
    big = tmp_path / "reads"
    big.mkdir()
    rng = np.random.default_rng(41)
    layouts = [(0, 0), (7, 1000), (3, 512), (16, 0), (19, 700), (32, 0), (32 | 16, 0), (128, 0), (256, 0), (64, 0), (32 | 64, 0), (512, 0), (1 | 2 | 4 | 8, 333)]
    for i in range(39):
        flags, chunk = layouts[i % len(layouts)]
        raw = synth_raw(rng, int(rng.integers(1500, 4200)))
        tmp = str(big / "x.i16")
        np.asarray(raw, dtype="<i2").tofile(tmp)
        subprocess.run([TOOL, "writex", str(big / ("read_%03d.fast5" % i)), "uuid-%04d" % i, "8192.0", "10.0", "1400.0", "4000.0", tmp, str(flags), str(chunk)], check=True)
        os.unlink(tmp)
    outs = {}
    for tag, extra, e2 in (("fast", ["--readers", "3"], {}), ("hdf5", ["--readers", "3"], {"FLAPPIE_DEBUG": "hdf5_read"}), ("fast_inproc", ["--readers", "0"], {})):
        r = subprocess.run([FLAPPIE, "--batch", "8"] + extra + [str(big)], env=dict(env, **e2), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs[tag] = r.stdout
    recs = _parse_fastq(outs["fast"])
    assert len(recs) == 39 and all(x[0].startswith("uuid-") for x in recs)          # the read_id attribute is the record's name (flappie.c:284-292 with --uuid, the default)
    assert outs["fast"] == outs["hdf5"] == outs["fast_inproc"]


@needs_hdf5
@pytest.mark.gpu
def test_batch_pipeline_across_chunks(cli_inputs, tmp_path):
    """the binary works in chunks of 4 x --batch reads and keeps its batch pipeline going across them (the first batch of a chunk
    is submitted before the last batch of the previous chunk is collected, a chunk is written while the next one runs): 150 reads
    through 6 chunks of tiny batches, through one chunk, and on the main thread only, give the same bytes in the same order --
    and --trace carries the same tables"""
    d, mdl, reads, raws = cli_inputs
    env = dict(os.environ, FLAPPIE_MODEL_DIR=str(d))
    big = tmp_path / "reads"
    big.mkdir()
    rng = np.random.default_rng(17)
    for i in range(150):
        write_fast5(big / ("read_%03d.fast5" % i), "uuid-%04d" % i, synth_raw(rng, int(rng.integers(1200, 3000))))
    outs = {}
    for tag, extra, e2 in (("chunks", ["--batch", "8", "--readers", "3"], {}), ("one", ["--batch", "256", "--readers", "2"], {}),
                           ("main", ["--batch", "8"], {"FLAPPIE_DEBUG": "no_reader_thread"})):
        r = subprocess.run([FLAPPIE, "--no-uuid", "--trace", str(tmp_path / (tag + ".hdf5"))] + extra + [str(big)], env=dict(env, **e2),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs[tag] = r.stdout
    assert len(_parse_fastq(outs["chunks"])) == 150
    assert outs["chunks"] == outs["one"] == outs["main"]
    # the next chunk's device pass begun AHEAD (ffhip_prep_begin / ffhip_prep_finish: the path a directory of long reads takes; prep_ahead_min=0 takes it with these small
    # chunks too) gives the same bytes, and the path was taken
    r = subprocess.run([FLAPPIE, "--no-uuid", "--batch", "8", "--readers", "3", str(big)], env=dict(env, FLAPPIE_DEBUG="prep_ahead_min=0,pack_log"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout == outs["chunks"]
    assert r.stderr.count("is there (") >= 1, r.stderr[-2000:]
    r = subprocess.run([FLAPPIE, "--no-uuid", "--batch", "8", "--readers", "3", str(big)], env=dict(env, FLAPPIE_DEBUG="no_prep_ahead"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout == outs["chunks"]
    for name in ("read_000.fast5", "read_077.fast5", "read_149.fast5"):
        sa, ta = dump_trace(tmp_path / "chunks.hdf5", name)
        sb, tb = dump_trace(tmp_path / "one.hdf5", name)
        assert np.array_equal(sa, sb) and np.array_equal(ta, tb) and ta.size > 0


@needs_hdf5
@pytest.mark.gpu
def test_pipeline_with_bad_chunks_a_dead_reader_and_a_closed_stdout(cli_inputs, tmp_path):
    """ADVICE r2, executed.  (a) More than two whole chunks of unreadable files in the middle of the input (--batch 8: a chunk is 32
    files) used to leave the batch in flight uncollected and its reader buffer unreleased: the run hung.  (b) A reader child that
    dies costs the file it died on; the rest of its stripe is read in-process and the exit status is non-zero.  (c) The binary
    dies of SIGPIPE like the reference when its stdout closes (`flappie ... | head`)."""
    d, mdl, reads, raws = cli_inputs
    env = dict(os.environ, FLAPPIE_MODEL_DIR=str(d))
    big = tmp_path / "reads"
    big.mkdir()
    rng = np.random.default_rng(23)
    good = []
    for i in range(130):
        fn = "read_%03d.fast5" % i
        if 20 <= i < 110:                                           # 90 bad files: chunks of 8, 32, 32, ... -> at least two all-bad chunks
            (big / fn).write_bytes(b"not an hdf5 file")
        else:
            write_fast5(big / fn, "uuid-%04d" % i, synth_raw(rng, int(rng.integers(1200, 2600))))
            good.append(fn)
    outs = {}
    for tag, extra, e2 in (("procs", ["--readers", "3"], {}), ("thread", ["--readers", "0"], {}), ("main", ["--readers", "0"], {"FLAPPIE_DEBUG": "no_reader_thread"})):
        r = subprocess.run([FLAPPIE, "--batch", "8", "--no-uuid"] + extra + [str(big)], env=dict(env, **e2), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert r.stderr.count("No basecall returned") == 90
        outs[tag] = r.stdout
    assert sorted(x[0] for x in _parse_fastq(outs["procs"])) == good
    assert outs["procs"] == outs["thread"] == outs["main"]
    # (b) reader 1 of 3 is killed when it reaches file 40 (of the sorted list): one read lost, everything else identical, status non-zero
    files = [str(big / fn) for fn in good]
    whole = subprocess.run([FLAPPIE, "--batch", "8", "--no-uuid", "--readers", "3"] + files, env=env, capture_output=True, text=True, timeout=300)
    assert whole.returncode == 0, whole.stderr
    r = subprocess.run([FLAPPIE, "--batch", "8", "--no-uuid", "--readers", "3"] + files, env=dict(env, FLAPPIE_DEBUG="kill_reader=1:20"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "ended early" in r.stderr and "reader process(es) failed" in r.stderr
    recs, ref = _parse_fastq(r.stdout), _parse_fastq(whole.stdout)
    lost = [x[0] for x in ref if x[0] not in {y[0] for y in recs}]
    assert len(lost) == 1 and good.index(lost[0]) % 3 == 1 and good.index(lost[0]) >= 20
    assert recs == [x for x in ref if x[0] != lost[0]]
    # (c) stdout closes after the first record: the process ends by SIGPIPE instead of basecalling everything into a closed pipe
    p = subprocess.Popen([FLAPPIE, "--batch", "8", "--no-uuid", "--readers", "2"] + files * 8, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    p.stdout.readline()
    p.stdout.close()
    assert p.wait(timeout=120) == -13


RELINKED = os.path.join(ROOT, "oracle", "_ref", "relink", "flappie_relinked")


@needs_hdf5
@pytest.mark.gpu
def test_reference_main_relinked_against_the_engine(cli_inputs):
    """INTEGRATION.md section 1, executed: the REFERENCE's own flappie.c (compiled unchanged by tools/relink_check.sh in the build
    container, where /root/reference lies, against include/ and linked with -lflappie_host -lffhip; the binary travels like the
    other oracle/_ref artefacts) basecalls fast5 files on the GPU through the reference-named functions -- one read per call,
    flappie.c:245-316 -- and prints the records the oracle expects, in the reference's own format."""
    if not os.path.exists(RELINKED):
        pytest.skip("oracle/_ref/relink/flappie_relinked not built (reference sources absent where build() ran)")
    d, mdl, reads, raws = cli_inputs
    env = dict(os.environ, FLAPPIE_MODEL_DIR=str(d), LD_LIBRARY_PATH=os.path.join(ROOT, "flappie_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    files = [str(reads / fn) for fn in sorted(raws)]
    r = subprocess.run([RELINKED, "--model", "r941_native"] + files, env=dict(env, FLAPPIE_REPORT_COPIES="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    recs = _parse_fastq(r.stdout)
    # the reference's main, one read per call: what crosses PCIe is the signal (up) and path / quality scores (down) -- never a matrix.
    # Per read: the signal + the convolution plans up; the abort word, path, qpath and score down.
    import re
    m = re.search(r"ffhip copies: h2d (\d+) calls (\d+) bytes, d2h (\d+) calls (\d+) bytes, largest d2h (\d+) bytes", r.stderr)
    assert m, r.stderr
    h2d_calls, h2d_bytes, d2h_calls, d2h_bytes, d2h_largest = (int(x) for x in m.groups())
    nread = len(files)
    nblock_max = max(len(raw) for _, raw in raws.values()) // 5 + 1
    # (flappie.c:299-300 computes the trace for every read, --trace or not: (nblock + 1) x 8 states of int32 is the largest thing that
    # comes down -- a fifth of a 40 x nblock matrix; the normalised signal of medmad_normalise_array, a host array in the reference's
    # API, comes down too)
    nsample_max = max(len(raw) for _, raw in raws.values())
    assert d2h_largest <= (nblock_max + 1) * 8 * 4 < nblock_max * 40 * 4
    assert d2h_calls <= 16 * nread + 8 and d2h_bytes <= nread * ((nblock_max + 1) * (8 + 32) + 4 * nsample_max + 256)
    assert h2d_bytes <= sum(len(raw) for _, raw in raws.values()) * 4 + nread * 8 * (nblock_max + 64) * 4 + (64 << 20)      # signals + per-read plan tables + the model, once
    ref = _oracle_calls(mdl, raws)
    assert [x[0] for x in recs] == [ref[fn]["uuid"] for fn in sorted(raws)]          # the reference's loop: argument order, uuid names
    for (name, hdr, bases, quals), fn in zip(recs, sorted(raws)):
        v = ref[fn]
        assert bases == v["basecall"] and quals == v["quality"], fn
        assert '"nblock" : %d' % v["nblock"] in hdr and '"trim" : [ %d, %d ]' % (v["start"], v["end"]) in hdr
    # and our own binary prints the same records for the same files: names, bases, qualities and every header field to the byte, except
    # "normalised_score" (minus the mean log-posterior along the path, %f).  The batch engine takes the posterior out of the SAME fp64 chains as logZ, on
    # exp(score - block max) of the un-normalised scores; transpost_crf_flipflop() here is handed the normalised fp32 scores and
    # exponentiates those -- the two differ by the fp32 rounding of "score - logZ / nblock", ~1e-7 in a log-posterior.
    ours = subprocess.run([FLAPPIE, "--model", "r941_native"] + files, env=env, capture_output=True, text=True, timeout=300)
    assert ours.returncode == 0
    score_re = re.compile(r'("normalised_score" : )(-?[0-9.]+|-?nan|-?inf)')
    mine = _parse_fastq(ours.stdout)
    assert len(mine) == len(recs)
    for a, b in zip(mine, recs):
        assert (a[0], a[2], a[3]) == (b[0], b[2], b[3])
        assert score_re.sub(r"\1S", a[1]) == score_re.sub(r"\1S", b[1])
        sa, sb = score_re.search(a[1]), score_re.search(b[1])
        assert sa and sb and abs(float(sa.group(2)) - float(sb.group(2))) <= 2e-6 * max(1.0, abs(float(sb.group(2)))), (a[1], b[1])


def test_shard_by_size_balances_bytes_and_partitions_the_list(tmp_path):
    """--shard g/n --shard-by-size (SURVEY.md section 8e "greedy by sum of samples"): every file lands in exactly one shard, the
    shards' byte sums are within one largest file of each other where dealing by index is not, and the table is flappie_amd/shard.py::
    partition_reads' (largest first, to the lightest shard).  No GPU: FLAPPIE_DEBUG=list_only prints the list and exits before the engine exists."""
    from flappie_amd import shard as S
    exe = os.path.join(ROOT, "flappie_amd", "flappie")
    if not os.path.exists(exe):
        pytest.fail("flappie binary not built")
    d = tmp_path / "reads"
    d.mkdir()
    rng = np.random.default_rng(8)
    sizes = {}
    for k in range(203):
        n = int(rng.integers(100, 5000)) * (40 if k % 8 == 0 else 1)          # every 8th file is fat: dealing by index puts them all in shard 0
        p = d / ("r%04d.fast5" % k)
        p.write_bytes(b"x" * n)
        sizes[str(p)] = n
    paths = sorted(sizes)
    env = dict(os.environ, FLAPPIE_DEBUG="list_only")

    def listing(extra):
        out = []
        for g in range(8):
            r = subprocess.run([exe, "--shard", "%d/8" % g] + extra + [str(d)], env=env, capture_output=True, text=True, timeout=60)
            assert r.returncode == 0, r.stderr
            out.append(r.stdout.split())
        return out

    by_index, by_size = listing([]), listing(["--shard-by-size"])
    for shards in (by_index, by_size):
        assert sorted(p for s in shards for p in s) == paths                       # a partition of the list
    want = S.partition_reads([sizes[p] for p in paths], 8)
    assert [[paths[i] for i in s] for s in want] == by_size
    load = lambda shards: [sum(sizes[p] for p in s) for s in shards]
    assert max(load(by_size)) - min(load(by_size)) <= max(sizes.values())
    assert max(load(by_index)) > 2 * min(load(by_index))
