"""The split-precision recurrent layer kernel (flappie_amd/csrc/ffhip_rnn_split.hip): recurrent layers of H = 128/256/384 run on
the 16-bit matrix pipes over a two-way fp16 split of both operands (three products per fp32 product, accuracy of an fp32
GEMM: ffhip_split.hpp, tests/test_split_numerics.py).  It is the default for those shapes, so the rest of the GPU suite (H = 36..96) never reaches it; these tests do, against the oracle
where the oracle is quick (H = 128) and against the f32-input MFMA kernel (FFHIP_RUN_F32_RNN) at the larger shapes.
Tolerances are those of the rest of the suite: 1e-4 on transition scores, identical base and quality strings."""
import numpy as np
import pytest

from flappie_amd import model as M
from oracle import ffo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from flappie_amd import binding
    return binding


@pytest.fixture(scope="module")
def engine(B):
    e = B.Engine(0)
    yield e
    e.close()


def check_read(b, r, ref):
    assert b.read_nblock(r) == ref["nblock"]
    dtrans = float(np.abs(b.transitions(r) - ref["trans"]).max())
    from conftest import note_parity
    note_parity(dtrans, np.abs(b.posterior(r) - ref["post"]).max())
    assert dtrans <= 5e-5          # half of north_star's 1e-4 (measured worst over the suite: 2.1e-5)
    path, _ = b.path(r)
    assert np.array_equal(path, ref["path"])
    assert b.basecall(r) == ref["basecall"] and b.quality(r) == ref["quality"]
    # end to end: the scores' own deviation propagates through two log-sum-exp recursions (measured worst 4.0e-5 at |dtrans| <= 2.1e-5);
    # the posterior kernel alone is held to 2e-5 + 2e-6 |x| on identical scores in tests/test_decode_gpu.py
    assert np.abs(b.posterior(r) - ref["post"]).max() <= 1e-4
    assert np.abs(b.trace(r) - ref["trace"]).max() <= 1


def test_split_kernel_against_oracle_odd_tile_count(B, engine):
    """40 reads = three read tiles: the second pair of tiles has one member only"""
    mdl = M.synthetic_model(M.NET_LSTM5, 128, seed=7)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(1623)
    sig = rng.standard_normal((40, 1500)).astype(np.float32)
    b = B.Batch(dm, 40, 1500)
    b.set_signals(sig)
    b.run(); b.finish()
    for r in range(40):
        check_read(b, r, om.basecall(sig[r]))
    b.close()
    dm.close()


def test_split_kernel_grumod_against_oracle(B, engine):
    """the GRUmod variant (r941_5mC family: gate rows z, r, candidate; the candidate's projection half rides in the free
    fourth row), uniform and ragged, against the oracle"""
    mdl = M.synthetic_model(M.NET_GRUMOD5, 128, seed=21)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(77)
    sig = rng.standard_normal((20, 1000)).astype(np.float32)
    b = B.Batch(dm, 20, 1000)
    b.set_signals(sig)
    b.run(); b.finish()
    assert b.rnn_path() == 3
    for r in range(20):
        check_read(b, r, om.basecall(sig[r]))
    lens = [1000, 999, 400, 37, 19, 640, 1000, 12 * 50, 333] + [0] * 8 + [777, 1000, 5]
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    lens[-1] = 0; sigs[-1] = np.zeros(0, dtype=np.float32)          # (5 samples would be shorter than the window)
    b.set_signals_ragged(sigs)
    b.run(); b.finish()
    for r, x in enumerate(sigs):
        if x.size:
            check_read(b, r, om.basecall(x))
    b.close()
    dm.close()


def test_split_kernel_ragged_and_empty_slots(B, engine):
    """per-read lengths, tiles whose block counts differ inside a pair, a tile of empty slots, and bitwise independence
    of a read's result from its slot and its neighbours"""
    mdl = M.synthetic_model(M.NET_LSTM5, 128, seed=11)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(5)
    lens = [1200, 1199, 600, 601, 37, 19, 1000, 800, 801, 802, 803, 804, 805, 806, 807, 808,       # tile 0
            300, 1200, 45]                                                                    # tile 1 (short) -> pair (0, 1)
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    b = B.Batch(dm, 40, 1200)                       # slots 19..39 stay empty: tile 2 is all empty, pair (2, -) never runs
    b.set_signals_ragged(sigs + [np.zeros(0, dtype=np.float32)] * (40 - len(sigs)))
    b.run(); b.finish()
    for r, x in enumerate(sigs):
        check_read(b, r, om.basecall(x))
    first = [b.transitions(r).copy() for r in range(len(sigs))]
    # the same reads in other slots (other tile of the pair, other neighbours): bit-identical per read
    order = list(reversed(range(len(sigs))))
    b.set_signals_ragged([np.zeros(0, dtype=np.float32)] * 17 + [sigs[k] for k in order] + [np.zeros(0, dtype=np.float32)] * (40 - 17 - len(sigs)))
    b.run(); b.finish()
    for slot, k in enumerate(order):
        assert np.array_equal(b.transitions(17 + slot), first[k]), k
    b.close()
    dm.close()


@pytest.mark.parametrize("kind,hidden,nread,T", [(M.NET_LSTM5, 256, 48, 2000), (M.NET_LSTM5, 384, 32, 1500), (M.NET_LSTM5, 128, 272, 500),
                                                 (M.NET_GRUMOD5, 256, 40, 1200), (M.NET_GRUMOD5, 384, 16, 800)])
def test_split_kernel_agrees_with_f32_kernel(B, engine, kind, hidden, nread, T):
    """H = 256 and 384 (two and three unit tiles per workgroup), LSTM and GRUmod; 272 reads = 17 read tiles, one more than a
    launch takes"""
    mdl = M.synthetic_model(kind, hidden, seed=hidden)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(nread)
    sig = rng.standard_normal((nread, T)).astype(np.float32)
    res = []
    for flags in (B.RUN_KEEP_ACTS, B.RUN_KEEP_ACTS | B.RUN_F32_RNN):
        b = B.Batch(dm, nread, T)
        b.set_signals(sig)
        b.run(1.0, flags); b.finish()
        res.append(([b.transitions(r) for r in range(nread)], [b.basecall(r) for r in range(nread)],
                    [b.quality(r) for r in range(nread)], [b.activation(4, r) for r in (0, nread - 1)]))
        b.close()
    assert max(float(np.abs(x - y).max()) for x, y in zip(res[0][0], res[1][0])) <= 1e-4
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    # last recurrent layer's output: what the fp32 copy written next to the split layout holds
    assert max(float(np.abs(x - y).max()) for x, y in zip(res[0][3], res[1][3])) <= 2e-5
    dm.close()


@pytest.mark.parametrize("kind,nread", [(M.NET_LSTM5, 512), (M.NET_GRUMOD5, 400), (M.NET_LSTM5, 768), (M.NET_GRUMOD5, 768), (M.NET_GRUMOD5, 1040), (M.NET_LSTM5, 1040),
                                        (M.NET_GRUMOD5, 2064), (M.NET_LSTM5, 2064)])
def test_dense_launch_matches_the_one_tile_launches(B, engine, kind, nread, monkeypatch):
    """H <= 256 and more than 256 reads: one launch carries up to 512 reads (pair form, two workgroups per CU) or -- round 3 -- 768
    (the dense form in 79 registers: THREE workgroups per CU) instead of launches of 256 -- same arithmetic, so every score must be
    IDENTICAL to what the one-tile launches give (FFHIP_DEBUG=no_dense); ragged lengths, the last pair of the 400-read batch has one
    member.
    From 1024 reads on: the PACKED forms (16 members a group, gate-major row tiles -- three a member for GRUmod, no empty accumulator rows, four
    for the LSTM; 1024 reads a launch) -- another tiling of the same sums in the same order, so identical as well; 1040 reads = one packed launch
    + 16 reads, 2064 = two + 16"""
    mdl = M.synthetic_model(kind, 256, seed=5 + kind)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(nread)
    T = 1000
    lens = rng.integers(200, T + 1, size=nread)
    lens[:3] = (T, 200, 237)
    sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
    res = []
    for dense in (True, False):
        if dense:
            monkeypatch.delenv("FFHIP_DEBUG", raising=False)
        else:
            monkeypatch.setenv("FFHIP_DEBUG", "no_dense")
        b = B.Batch(dm, nread, T)
        b.set_signals_ragged(sigs)
        b.run(); b.finish()
        res.append([(b.transitions(r), b.basecall(r), b.quality(r)) for r in range(nread)])
        b.close()
    for r in range(nread):
        assert np.array_equal(res[0][r][0], res[1][r][0]), r
        assert res[0][r][1:] == res[1][r][1:], r
    dm.close()


# (384 x 512 and 256 x 768: every layer launch fills the chip -- such batches run one after the other on the GPU, ffhip_engine.hip "whole batches")
@pytest.mark.parametrize("kind,hidden,nread", [(M.NET_LSTM5, 384, 256), (M.NET_LSTM5, 256, 512), (M.NET_GRUMOD5, 256, 256), (M.NET_LSTM5, 384, 512), (M.NET_LSTM5, 256, 768), (M.NET_GRUMOD5, 256, 1024), (M.NET_LSTM5, 256, 1024)])
def test_two_batches_in_flight_give_the_results_of_one(B, engine, kind, hidden, nread):
    """bench.py's default at c2 and the flappie binary keep two batches in flight (one stream each; the persistent layer launches of
    the two are chained, every other kernel overlaps the other batch's layers).  Three rounds of run / run / finish / finish with
    different signals per batch: every score identical to the same batch run alone"""
    mdl = M.synthetic_model(kind, hidden, seed=3)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(hidden + nread)
    T = 1500
    sig = [rng.standard_normal((nread, T)).astype(np.float32) for _ in range(2)]
    alone = []
    for k in range(2):
        b = B.Batch(dm, nread, T)
        b.set_signals(sig[k])
        b.run(); b.finish()
        alone.append([(b.transitions(r), b.basecall(r), b.quality(r)) for r in range(0, nread, 7)])
        b.close()
    bs = [B.Batch(dm, nread, T) for _ in range(2)]
    for rnd in range(3):
        for k in range(2):
            bs[k].set_signals(sig[(k + rnd) % 2])
            bs[k].run()
        for k in range(2):
            bs[k].finish()
            want = alone[(k + rnd) % 2]
            for i, r in enumerate(range(0, nread, 7)):
                assert np.array_equal(bs[k].transitions(r), want[i][0]), (rnd, k, r)
                assert (bs[k].basecall(r), bs[k].quality(r)) == want[i][1:], (rnd, k, r)
    for b in bs:
        b.close()
    dm.close()


@pytest.mark.parametrize("kind", [M.NET_LSTM5, M.NET_GRUMOD5])
def test_packed_kernels_against_oracle(B, engine, kind):
    """the packed layer kernels of H = 256 (k_lstm_pack / k_grumod_pack: full 1024-read launches only) against the oracle itself: a batch
    of 1024 slots of which 28 hold a read (both tiles of a group, one tile only, the first and the last slot; whole groups empty), so that the
    oracle's share stays a few seconds; profiles/r03_parity_packed.txt is the same on 2048 reads"""
    mdl = M.synthetic_model(kind, 256, seed=3)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    assert dm.launch_reads == 1024
    rng = np.random.default_rng(404 + kind)
    slots = [0, 1, 15, 16, 17, 31, 40, 100, 101, 250, 255, 256, 300, 511, 512, 513, 600, 640, 655, 700, 767, 768, 900, 990, 1000, 1008, 1022, 1023]
    sigs = [np.zeros(0, dtype=np.float32)] * 1024
    for k in slots:
        sigs[k] = rng.standard_normal(int(rng.integers(150, 700))).astype(np.float32)
    sigs[0] = rng.standard_normal(700).astype(np.float32)
    b = B.Batch(dm, 1024, 700)
    b.set_signals_ragged(sigs)
    b.run(); b.finish()
    assert b.rnn_path() == 3
    for k in slots:
        check_read(b, k, om.basecall(sigs[k]))
    b.close()
    dm.close()


def test_launch_reads_is_what_a_full_layer_launch_takes(B, engine):
    """ffhip_model_launch_reads (the flappie binary's default batch size): per 32 CUs 8 read tiles at H = 256 (the packed forms), 4 at
    H = 384, 2 at H = 512 and for the shapes outside the split kernels"""
    unit = engine.info()["ncu"] // 32
    for kind, hidden, tiles in ((M.NET_GRUMOD5, 256, 8), (M.NET_LSTM5, 256, 8), (M.NET_LSTM5, 384, 4), (M.NET_LSTM5, 512, 2), (M.NET_LSTM5, 96, 2)):
        dm = B.DeviceModel(engine, M.synthetic_model(kind, hidden, seed=1))
        assert dm.launch_reads == 16 * tiles * unit, (kind, hidden)
        dm.close()


def test_paired_layer_launches_give_the_results_of_each_batch_alone(B, engine):
    """ffhip_batch_run_pair: at H = 384 the recurrent layers of two 256-read batches are ONE launch per layer (k_lstm_split_pair, the
    dense form: two workgroups per CU).  Uniform and ragged pairs over three rounds, two pairs in flight as bench.py keeps them: every
    score, base and quality string identical to the same batch run alone through ffhip_batch_run; shapes the paired launch does not
    take (another hidden size, fewer reads) go through the same call and run one after the other."""
    rng = np.random.default_rng(384)
    for hidden, nread, T, ragged in ((384, 256, 1200, False), (384, 256, 1500, True), (256, 64, 900, False), (384, 48, 900, False)):
        mdl = M.synthetic_model(M.NET_LSTM5, hidden, seed=5)
        dm = B.DeviceModel(engine, mdl)

        def signals():
            if not ragged:
                return rng.standard_normal((nread, T)).astype(np.float32)
            lens = rng.integers(T // 2, T + 1, nread)
            lens[rng.random(nread) < 0.1] = 0
            lens[0] = T
            return [rng.standard_normal(int(n)).astype(np.float32) for n in lens]

        def load(b, sg):
            b.set_signals_ragged(sg) if ragged else b.set_signals(sg)

        def live(sg, r):
            return len(sg[r]) > 0

        sigs = [signals() for _ in range(4)]
        probe = list(range(0, nread, 11))
        alone = []
        for sg in sigs:
            b = B.Batch(dm, nread, T)
            load(b, sg)
            b.run(); b.finish()
            assert not b.paired()
            alone.append({r: (b.transitions(r), b.basecall(r), b.quality(r)) for r in probe if live(sg, r)})
            b.close()
        bs = [B.Batch(dm, nread, T) for _ in range(4)]
        for rnd in range(3):
            order = [(k + rnd) % 4 for k in range(4)]
            for k in range(4):
                load(bs[k], sigs[order[k]])
            bs[0].run_pair(bs[1]); bs[2].run_pair(bs[3])          # two pairs in flight
            for k in range(4):
                bs[k].finish()
                assert bs[k].paired() == (hidden == 384 and nread == 256), (hidden, nread)
                for r, want in alone[order[k]].items():
                    assert np.array_equal(bs[k].transitions(r), want[0]), (hidden, nread, ragged, rnd, k, r)
                    assert (bs[k].basecall(r), bs[k].quality(r)) == want[1:], (hidden, nread, ragged, rnd, k, r)
        for b in bs:
            b.close()
        dm.close()


def test_packed_fp32_forms_beside_the_layer_kernel(B, engine):
    """beside a wave that issues 16-bit MFMAs, v_pk_{add,mul,fma}_f32 ... op_sel:[0,1] returns a wrong low
    half in lanes 48-63; tools/check_isa.py keeps that form out of the library.  The guard is only as good as its list, so: the
    probe (every op_sel form against the scalar instructions) must be clean alone, and beside an H = 256 batch's layer kernels
    every OTHER form must stay clean -- a new failing form has to show up here."""
    import ctypes as C
    lib = B.lib()
    lib.ffhip_debug_pk_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint)]

    def probe():
        buf = (C.c_uint * (4 * 16 * 2 * 4))()
        assert lib.ffhip_debug_pk_probe(engine.h, 4000, 2048, 1, buf) == 0
        return np.frombuffer(buf, dtype=np.uint32).reshape(4, 16, 2, 4).copy()

    assert probe().sum() == 0
    mdl = M.synthetic_model(M.NET_LSTM5, 256, seed=3)
    dm = B.DeviceModel(engine, mdl)
    b = B.Batch(dm, 256, 20000)
    b.set_signals(np.random.default_rng(9).standard_normal((256, 20000)).astype(np.float32))
    b.run(1.0, B.RUN_NO_DECODE)
    c = probe()
    b.finish()
    b.close(); dm.close()
    known = np.zeros_like(c, dtype=bool)
    known[0:3, 4:8, 0, :] = True                 # add / mul / fma, op_sel:[0,1] with any op_sel_hi, low half
    assert c[~known].sum() == 0, np.argwhere((c > 0) & ~known)[:8]
    print("packed-fp32 probe beside the H = 256 layer kernels: %d mismatches, all in op_sel:[0,1] low halves; by wave quarter %s"
          % (int(c.sum()), c[known].reshape(-1, 4).sum(axis=0).tolist()))


def test_first_ragged_batch_of_a_new_batch_object_beside_another(B, engine):
    """regression: the tables of a ragged batch go up asynchronously on the batch's stream, and a batch object allocates (and
    zero-fills) some of them on first ragged use -- a zero-fill on the NULL stream could overtake the upload and wipe the per-read
    block counts (seen by tools/stress.py as a wrong first ragged batch of the second batch object on most boxes).  Eight fresh
    pairs of batch objects, H = 512 and GRUmod H = 256: a probe read in the ragged batch must come out as in a uniform one"""
    rng = np.random.default_rng(3)
    probe = rng.standard_normal(2777).astype(np.float32)
    for kind, hidden, nread in ((M.NET_LSTM5, 512, 256), (M.NET_GRUMOD5, 256, 512)):
        mdl = M.synthetic_model(kind, hidden, seed=1)
        dm = B.DeviceModel(engine, mdl)
        ref = None
        for rep in range(4):
            b0, b1 = B.Batch(dm, nread, 4000), B.Batch(dm, nread, 4000)
            uni = [probe if i == 7 else rng.standard_normal(4000).astype(np.float32) for i in range(nread)]
            lens = np.sort(rng.integers(1500, 4000, nread))[::-1].copy()
            slot = int(rng.integers(0, nread))
            lens[slot] = probe.size
            rag = [probe if i == slot else rng.standard_normal(int(n)).astype(np.float32) for i, n in enumerate(lens)]
            b0.set_signals_ragged(uni); b0.run()
            b1.set_signals_ragged(rag); b1.run()              # first (ragged) use of b1, beside b0
            b0.finish(); b1.finish()
            got0 = (b0.basecall(7), b0.quality(7), b0.transitions(7).tobytes())
            got1 = (b1.basecall(slot), b1.quality(slot), b1.transitions(slot).tobytes())
            ref = ref or got0
            assert got0 == ref and got1 == ref, (kind, hidden, rep)
            b0.close(); b1.close()
        dm.close()


@pytest.mark.parametrize("kind,hidden,nread", [(0, 384, 256), (1, 256, 512)])
def test_short_soak_two_batches_in_flight(B, kind, hidden, nread):
    """a minute's worth of tools/stress.py in a few seconds: 30 uniform / sorted-ragged / unsorted-ragged batches through two batch
    objects in flight, a probe read in a random slot of each; its bases, qualities, scores and path must never change"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress.py"), "30", str(kind), str(hidden), "2", str(nread)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "stress ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_split_layout_round_trip(B, engine):
    """fp32 -> slices -> fp32 through the operand format of the split layer kernels (ffhip_split.hpp).  Default build: two
    fp16 slices of x * 2^12 hold |x| <= 1 to 2^-22 relative (absolute floor 2^-37); the -DFFHIP_SPLIT_BF16X3 build's three
    bf16 slices are the identity on every fp32."""
    import ctypes as C
    L = B.lib()
    if not hasattr(L, "ffhip_debug_split_round_trip"):
        pytest.skip("debug entry point not built")
    rng = np.random.default_rng(0)
    x = np.tanh(rng.standard_normal(16 * 128 * 3)).astype(np.float32)
    x[1000:2000] *= np.float32(1e-3)
    x[2000:3000] *= np.float32(1e-6)
    x[:8] = np.float32([0.0, 1.0, -1.0, 1e-20, 0.99999994, -0.99999994, 1.17549435e-38, 0.5])
    y = np.empty_like(x)
    L.ffhip_debug_split_round_trip.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t, C.c_int]
    L.ffhip_debug_split_round_trip.restype = C.c_int
    assert L.ffhip_debug_split_round_trip(engine.h, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)), 3, 128) == 0
    err = np.abs(x.astype(np.float64) - y.astype(np.float64))
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -36)), float((err / np.maximum(np.abs(x), 1e-30)).max())
    assert np.array_equal(y[:3], x[:3])


def test_lean_gate_math_is_bit_identical(B, engine):
    """the gate phase of the layer kernels evaluates 1 / (1 + exp(-x)) with a Newton reciprocal and floor() where the
    reference-order code divides and truncates/compares/subtracts (ffhip_math.hpp, *_lean): every fp32 mantissa at
    several binary exponents must give the same bits, for the reciprocal, the logistic and tanh"""
    import ctypes as C
    L = B.lib()
    L.ffhip_debug_lean_math_check.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    L.ffhip_debug_lean_math_check.restype = C.c_int
    for ex in (0, 1, 2, 3, 4, 5, 6, 7, 23, 100, 125):
        n = C.c_ulonglong(1)
        assert L.ffhip_debug_lean_math_check(engine.h, ex, 1, C.byref(n)) == 0
        assert n.value == 0, ex


@pytest.mark.parametrize("kind", [M.NET_LSTM5, M.NET_GRUMOD5])
def test_split_kernel_very_short_reads(B, engine, kind):
    """batches whose reads have fewer blocks than the kernel looks ahead (sentinels three steps, L2 warming three steps, the
    projection one step): 1 to 7 blocks"""
    mdl = M.synthetic_model(kind, 128, seed=31)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(3)
    for T in (19, 20, 24, 26, 31, 37):
        sig = rng.standard_normal((18, T)).astype(np.float32)
        b = B.Batch(dm, 18, T)
        b.set_signals(sig)
        b.run(); b.finish()
        assert b.rnn_path() == 3
        for r in (0, 15, 16, 17):
            check_read(b, r, om.basecall(sig[r]))
        b.close()
    dm.close()


@pytest.mark.parametrize("kind", [M.NET_LSTM5, M.NET_GRUMOD5])
def test_split_projection_gemm_on_the_unfused_path(B, engine, kind):
    """FFHIP_RUN_UNFUSED_RNN at H % 128 == 0 (what H = 512 models always take): the input projection is a GEMM on split
    operands in front of the f32 persistent recurrence"""
    mdl = M.synthetic_model(kind, 128, seed=41)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(8)
    sig = rng.standard_normal((20, 900)).astype(np.float32)
    b = B.Batch(dm, 20, 900)
    b.set_signals(sig)
    b.run(1.0, B.RUN_UNFUSED_RNN); b.finish()
    assert b.rnn_path() == 1
    for r in (0, 7, 16, 19):
        check_read(b, r, om.basecall(sig[r]))
    b.close()
    dm.close()


def test_recurrence_only_split_kernel(B, engine):
    """the H = 512 arrangement (projection GEMM on split operands + recurrence-only split layer kernel, rnn_path 4), exercised
    at H = 256 where the oracle is quick: uniform with an odd tile count, then ragged with empty slots"""
    mdl = M.synthetic_model(M.NET_LSTM5, 256, seed=51)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(12)
    sig = rng.standard_normal((40, 700)).astype(np.float32)
    b = B.Batch(dm, 40, 700)
    b.set_signals(sig)
    b.run(1.0, B.RUN_UNFUSED_RNN); b.finish()
    assert b.rnn_path() == 4
    for r in (0, 15, 16, 31, 32, 39):
        check_read(b, r, om.basecall(sig[r]))
    lens = [700, 699, 300, 37, 19, 640] + [0] * 11 + [555, 700, 21] + [0] * 20
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    b.set_signals_ragged(sigs)
    b.run(1.0, B.RUN_UNFUSED_RNN); b.finish()
    for r, x in enumerate(sigs):
        if x.size:
            check_read(b, r, om.basecall(x))
    b.close()
    dm.close()


def test_differential_fuzz_is_deterministic(B):
    """a few seconds of tools/dev/diff_fuzz.py: random shapes through the split path twice and the f32 path once; the script
    asserts bit-for-bit determinism of the split path (differing strings between the two paths are reported, not asserted:
    DESIGN.md section 3)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "dev", "diff_fuzz.py"), "6", "7"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "diff fuzz:" in out.stdout


def test_timeout_falls_back_to_the_stepwise_kernels(B):
    """Co-residency guard: when a persistent layer kernel gives up waiting for peer workgroups -- another
    tenant holds part of the GPU -- ffhip_batch_finish re-runs the batch on the launch-per-step kernels instead of returning
    FFHIP_ETIMEOUT, and the next runs go there directly.  The time-out is simulated (FFHIP_DEBUG=force_abort pre-sets the abort
    word; the layer kernels then leave at once, as they do when an earlier layer of a batch timed out)."""
    import ctypes as C
    import os
    eng = B.Engine(0)                       # its own engine: the fallback state is per engine
    L = B.lib()
    L.ffhip_debug_fallback_count.argtypes = [C.c_void_p]
    L.ffhip_debug_fallback_count.restype = C.c_int
    mdl = M.synthetic_model(M.NET_LSTM5, 128, seed=7)
    dm = B.DeviceModel(eng, mdl)
    sig = np.random.default_rng(5).standard_normal((20, 900)).astype(np.float32)
    b = B.Batch(dm, 20, 900)
    b.set_signals(sig)
    b.run(); b.finish()
    assert b.rnn_path() == 3 and L.ffhip_debug_fallback_count(eng.h) == 0
    want = [(b.basecall(r), b.quality(r), b.transitions(r)) for r in range(20)]
    os.environ["FFHIP_DEBUG"] = "force_abort"
    try:
        b.run(); b.finish()                 # "times out", is re-run stepwise inside finish()
    finally:
        del os.environ["FFHIP_DEBUG"]
    assert L.ffhip_debug_fallback_count(eng.h) == 1 and b.rnn_path() == 0
    for r in range(20):
        assert b.basecall(r) == want[r][0] and b.quality(r) == want[r][1]
        assert np.abs(b.transitions(r) - want[r][2]).max() <= 1e-4
    b.run(); b.finish()                     # still wary of the co-tenant: straight to the stepwise kernels, no second fallback
    assert b.rnn_path() == 0 and L.ffhip_debug_fallback_count(eng.h) == 1
    os.environ["FFHIP_NO_FALLBACK"] = "1"
    os.environ["FFHIP_DEBUG"] = "force_abort"
    try:
        eng2 = B.Engine(0)
        dm2 = B.DeviceModel(eng2, mdl)
        b2 = B.Batch(dm2, 20, 900)
        b2.set_signals(sig)
        b2.run()
        with pytest.raises(B.FFHipError):   # the old behaviour stays available
            b2.finish()
        b2.close(); dm2.close(); eng2.close()
    finally:
        del os.environ["FFHIP_NO_FALLBACK"], os.environ["FFHIP_DEBUG"]
    b.close(); dm.close(); eng.close()


def test_outlier_samples_beyond_the_split_format_are_run_again_on_the_f32_path(B, engine):
    """The split operand format carries the swish convolutions' outputs as fp16 slices of value * 16, i.e. up to +-4094 (ffhip_split.hpp); the
    reference's swish has no bound (layers.c:24-33).  A med-MAD normalised signal stays orders of magnitude below that, but nothing is returned
    clamped: the kernels that produce the format flag a read that passes the bound, and ffhip_batch_finish runs it again
    through the all-f32 kernels and puts those results in place.  (a) outliers inside the range change nothing and no read is re-run; (b) a
    spike that drives the second convolution beyond 4094: the DEFAULT path returns the oracle's call -- bit for bit what FFHIP_RUN_F32_RNN
    gives for that read -- and counts one re-run; (c) in a ragged 40-read batch with three such reads, the other 37 are bit-identical to
    the same batch without the spikes; (d) the same through ffhip_batch_run_pair."""
    from oracle import ffo
    mdl = M.synthetic_model(M.NET_LSTM5, 128, seed=9)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(77)
    base = rng.standard_normal(1500).astype(np.float32)
    mild, wild = base.copy(), base.copy()
    mild[[300, 301, 900]] = [60.0, -45.0, 80.0]                 # large for a normalised signal, far inside the format
    wild[[300, 301, 900]] = [6.0e4, -4.5e4, 8.0e4]              # drives the second convolution's output beyond 4094
    res = {}
    before = engine.f32_reruns()
    for name, sig in (("mild", mild), ("wild", wild)):
        ref = om.basecall(sig)
        for tag, flags in (("split", 0), ("f32", B.RUN_F32_RNN)):
            b = B.Batch(dm, 1, sig.size)
            b.set_signals(sig[None, :])
            b.run(1.0, flags); b.finish()
            tr = b.transitions(0)
            assert np.isfinite(tr).all()
            res[(name, tag)] = (float(np.abs(tr - ref["trans"]).max()), b.basecall(0) == ref["basecall"] and b.quality(0) == ref["quality"], b.f32_reruns(), tr,
                                np.array_equal(b.path(0)[0], ref["path"]), float(np.abs(b.posterior(0) - ref["post"]).max()), int(np.abs(b.trace(0) - ref["trace"]).max()))
            b.close()
    print("max |dtrans| vs oracle, strings equal, reads re-run:", {k: v[:3] for k, v in res.items()})
    assert res[("mild", "split")][0] <= 1e-4 and res[("mild", "split")][1] and res[("mild", "split")][2] == 0
    assert res[("mild", "f32")][0] <= 1e-4 and res[("mild", "f32")][1]
    # with activations of 1e4 behind the spikes the summation orders differ by more than on ordinary reads (measured 1.7e-4)
    assert res[("wild", "f32")][0] <= 5e-4 and res[("wild", "f32")][1] and res[("wild", "f32")][2] == 0
    w = res[("wild", "split")]
    assert w[2] == 1 and engine.f32_reruns() == before + 1
    assert np.array_equal(w[3], res[("wild", "f32")][3])                      # the default path's answer IS the f32 path's
    assert w[0] <= 5e-4 and w[1] and w[4] and w[5] <= 1e-3 and w[6] <= 1     # ... and the oracle's call
    # (c) a ragged batch: three of 40 reads carry a spike
    lens = rng.integers(400, 1501, 40)
    clean = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
    spiked = [x.copy() for x in clean]
    for r, at in ((3, 200), (16, 50), (39, 390)):
        spiked[r][at] = 7.0e4
    out = []
    for sg in (clean, spiked):
        b = B.Batch(dm, 40, 1500)
        b.set_signals_ragged(sg)
        b.run(); b.finish()
        out.append(([b.transitions(r) for r in range(40)], [(b.basecall(r), b.quality(r), b.score(r)) for r in range(40)], b.f32_reruns()))
        if sg is spiked:
            for r in (3, 16, 39):
                ref = om.basecall(sg[r])
                assert float(np.abs(b.transitions(r) - ref["trans"]).max()) <= 5e-4
                assert (b.basecall(r), b.quality(r)) == (ref["basecall"], ref["quality"]) and np.array_equal(b.path(r)[0], ref["path"])
                assert np.abs(b.trace(r) - ref["trace"]).max() <= 1
        b.close()
    assert out[0][2] == 0 and out[1][2] == 3
    for r in range(40):
        if r not in (3, 16, 39):
            assert np.array_equal(out[0][0][r], out[1][0][r]) and out[0][1][r] == out[1][1][r], r
    dm.close()
    # (d) the paired launch of bench.py's shape: one read of the second batch carries a spike
    mdl = M.synthetic_model(M.NET_LSTM5, 384, seed=1)
    dm = B.DeviceModel(engine, mdl)
    sig = [rng.standard_normal((256, 600)).astype(np.float32) for _ in range(2)]
    sig[1][77, 300] = -9.0e4
    bs = [B.Batch(dm, 256, 600) for _ in range(2)]
    for b, sg in zip(bs, sig):
        b.set_signals(sg)
    bs[0].run_pair(bs[1])
    for b in bs:
        b.finish()
        assert b.paired()
    assert bs[0].f32_reruns() == 0 and bs[1].f32_reruns() == 1
    ref = ffo.OracleModel(mdl).basecall(sig[1][77])
    assert float(np.abs(bs[1].transitions(77) - ref["trans"]).max()) <= 5e-4 and bs[1].basecall(77) == ref["basecall"] and bs[1].quality(77) == ref["quality"]
    ref = ffo.OracleModel(mdl).basecall(sig[1][78])
    assert float(np.abs(bs[1].transitions(78) - ref["trans"]).max()) <= 1e-4 and bs[1].basecall(78) == ref["basecall"]
    for b in bs:
        b.close()
    dm.close()
