"""Ragged batches: reads of different lengths in one batch, each evaluated whole and exactly as if alone
(include/ffhip.h, ffhip_batch_set_reads).  Real reads never share a length, so this is what makes the batched
engine usable on real data without padding, chunking or stitching (none of which the reference does)."""
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


def check_read(b, r, ref, viterbi_only=False):
    assert b.read_nblock(r) == ref["nblock"]
    dtrans = float(np.abs(b.transitions(r) - ref["trans"]).max())
    from conftest import note_parity
    note_parity(dtrans, None if viterbi_only else np.abs(b.posterior(r) - ref["post"]).max())
    assert dtrans <= 5e-5          # half of north_star's 1e-4 (measured worst over the suite: 2.1e-5)
    path, qpath = b.path(r)
    assert np.array_equal(path, ref["path"])
    assert b.basecall(r) == ref["basecall"] and b.quality(r) == ref["quality"]
    assert abs(b.score(r) - ref["score"]) <= 1e-3 * max(1.0, abs(ref["score"]))
    if not viterbi_only:
        # end to end: the scores' own deviation propagates through two log-sum-exp recursions (measured worst 4.0e-5 at |dtrans| <= 2.1e-5);
        # the posterior kernel alone is held to 2e-5 + 2e-6 |x| on identical scores in tests/test_decode_gpu.py
        assert np.abs(b.posterior(r) - ref["post"]).max() <= 1e-4
        assert np.abs(b.trace(r) - ref["trace"]).max() <= 1


@pytest.mark.parametrize("kind,hidden,lens", [
    (M.NET_LSTM5, 64, [2000, 1999, 1996, 1500, 1003, 777, 100, 2000, 1234, 1235, 1236, 1237, 1238, 501, 19, 45, 1600, 23]),
    (M.NET_GRUMOD5, 48, [1200, 1199, 600, 601, 37, 19, 1000]),
    (M.NET_LSTM5, 96, [900, 500]),
])
def test_ragged_batch_equals_each_read_alone(B, engine, kind, hidden, lens):
    """every right-edge case of the strided convolution (lengths mod stride), reads shorter than one tile's
    neighbours, a second partly filled tile, the shortest legal read (19 samples = the window)"""
    mdl = M.synthetic_model(kind, hidden, seed=17)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(len(lens))
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    b = B.Batch(dm, len(lens), max(lens))
    for viterbi_only in (False, True):
        b.set_signals_ragged(sigs)
        b.run(1.0, B.RUN_VITERBI_ONLY if viterbi_only else 0)
        b.finish()
        for r, x in enumerate(sigs):
            check_read(b, r, om.basecall(x, viterbi_only=viterbi_only), viterbi_only)
    # the same batch object back on the uniform path, then ragged again with other lengths
    uni = rng.standard_normal((len(lens), max(lens))).astype(np.float32)
    b.set_signals(uni)
    b.run(); b.finish()
    for r in (0, len(lens) - 1):
        check_read(b, r, om.basecall(uni[r]))
    sigs2 = [rng.standard_normal(n).astype(np.float32) for n in reversed(lens)]
    b.set_signals_ragged(sigs2)
    b.run(); b.finish()
    for r in (0, 1, len(lens) - 1):
        check_read(b, r, om.basecall(sigs2[r]))
    b.close()
    dm.close()


def test_ragged_variants_and_errors(B, engine):
    """the three recurrent implementations mask the same way; lengths outside the domain are refused"""
    mdl = M.synthetic_model(M.NET_LSTM5, 96, seed=3)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(1)
    lens = [1500, 1497, 900, 1499, 333, 1500, 64, 1100, 1101, 1102, 1103, 1104, 1105, 1106, 1107, 1108, 700, 300, 1500, 21]
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    outs = []
    for flags in (0, B.RUN_UNFUSED_RNN, B.RUN_STEPWISE_RNN):
        b = B.Batch(dm, len(lens), 1500)
        b.set_signals_ragged(sigs)
        b.run(1.0, flags); b.finish()
        outs.append(([b.transitions(r) for r in (0, 4, 6, 19)], [b.basecall(r) for r in range(len(lens))]))
        b.close()
    for tr, calls in outs[1:]:
        for a, c in zip(outs[0][0], tr):
            assert a.shape == c.shape and np.abs(a - c).max() <= 5e-5
        assert calls == outs[0][1]
    b = B.Batch(dm, 2, 500)
    with pytest.raises(B.FFHipError):
        b.set_signals_ragged([rng.standard_normal(500).astype(np.float32), rng.standard_normal(18).astype(np.float32)])    # shorter than the window
    with pytest.raises(B.FFHipError):
        b.set_signals_ragged([rng.standard_normal(501).astype(np.float32), rng.standard_normal(100).astype(np.float32)])   # beyond capacity
    b.close()
    dm.close()


def test_ragged_runlength_and_prepared(B, engine):
    mdl = M.synthetic_model(M.NET_LSTM5_RLE, 48, seed=4)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(6)
    lens = [800, 640, 799, 100]
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    b = B.Batch(dm, len(lens), 800)
    b.set_signals_ragged(sigs)
    b.run(); b.finish()
    for r, x in enumerate(sigs):
        ref = om.runlength_call(x)
        path, _ = b.path(r)
        assert np.array_equal(path[:-1], ref["path"]) and np.abs(b.transitions(r) - ref["param"]).max() <= 1e-4
    b.close(); dm.close()
    # GPU-prepared raw reads of different trimmed lengths straight into one batch
    mdl = M.synthetic_model(M.NET_LSTM5, 48, seed=5)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    from test_signal_prep_gpu import oracle_prep, synth_raw
    raws = [synth_raw(rng, n, lead=l) for n, l in ((4000, 600), (3100, 0), (2777, 250), (3999, 100), (1500, 0))]
    p = B.Prepared(engine, raws)
    kept = [p.range(i)[1] - p.range(i)[0] for i in range(len(raws))]
    assert len(set(kept)) > 1
    order = sorted(range(len(raws)), key=lambda i: -kept[i])
    b = B.Batch(dm, len(raws), max(kept))
    b.set_prepared(p, order)
    b.run(); b.finish()
    for k, i in enumerate(order):
        ref = om.basecall(oracle_prep(raws[i])[2])
        assert b.basecall(k) == ref["basecall"] and b.quality(k) == ref["quality"]
    b.close(); p.close(); dm.close()


def test_empty_slots(B, engine):
    """slots left empty (length 0 / prepared index -1) cost nothing and leave the other reads' results untouched: one
    batch object serves groups of any size up to its own"""
    mdl = M.synthetic_model(M.NET_LSTM5, 64, seed=11)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(2)
    nslot = 40                                                   # three read tiles, the last two mostly / wholly empty
    lens = [900, 0, 650, 0, 0, 901] + [0] * 12 + [333] + [0] * 21
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    b = B.Batch(dm, nslot, 1000)
    b.set_signals_ragged(sigs)
    b.run(); b.finish()
    for r, x in enumerate(sigs):
        if x.size:
            check_read(b, r, om.basecall(x))
        else:
            assert b.read_nblock(r) == 0
            with pytest.raises(B.FFHipError):
                b.transitions(r)
    b.close(); dm.close()


def test_leading_empty_slots(B, engine):
    """an empty FIRST slot / first read tile: the decode kernels of an empty read must not touch block Tb - 1 = -1,
    which for read 0 lies in front of the buffers"""
    mdl = M.synthetic_model(M.NET_LSTM5, 64, seed=5)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(9)
    lens = [0] * 17 + [700, 699, 0, 19, 400]
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    b = B.Batch(dm, len(lens), 700)
    for flags in (0, B.RUN_VITERBI_ONLY):
        b.set_signals_ragged(sigs)
        b.run(1.0, flags); b.finish()
        for r, x in enumerate(sigs):
            if x.size:
                check_read(b, r, om.basecall(x, viterbi_only=bool(flags)), bool(flags))
            else:
                assert b.read_nblock(r) == 0
    b.close(); dm.close()
