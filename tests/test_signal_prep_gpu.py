"""GPU signal preparation (ffhip_prep.hip: SURVEY.md section 8f row N1) -- trim_and_segment_raw and
medmad_normalise_array / --delta as exact selections on the device.  This is the part of the path the
reference's own test fixtures pin (test_flappie_signal.c:67-111: raw_signal.crp -> trimmed_signal.crp ->
normalised_signal.crp), so the GPU result is checked against those fixtures first, then bit-for-bit against
the oracle (which tests/test_oracle_cpu.py checks against the compiled reference sources)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import ffo

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


@pytest.fixture(scope="module")
def B():
    from flappie_amd import binding
    return binding


@pytest.fixture(scope="module")
def engine(B):
    e = B.Engine(0)
    yield e
    e.close()


def oracle_prep(raw, trim_start=200, trim_end=10, chunk=100, perc=0.0):
    """(start, end, normalised signal) from the oracle; None if the read is rejected"""
    x = np.ascontiguousarray(raw, dtype=np.float32)
    s, e = C.c_size_t(0), C.c_size_t(x.size)
    rc = ffo.lib().fo_trim_and_segment_raw(_f(x), x.size, C.byref(s), C.byref(e), trim_start, trim_end, chunk, perc)
    if rc != 0:
        return None
    y = x[s.value:e.value].copy()
    ffo.lib().fo_medmad_normalise_array(_f(y), y.size)
    return s.value, e.value, y


def synth_raw(rng, n, lead=0, tail=0):
    x = rng.normal(90, 12, n).astype(np.float32)
    if lead:
        x[:lead] = rng.normal(95, 0.8, lead)
    if tail:
        x[n - tail:] = rng.normal(60, 0.5, tail)
    return np.round(x * 8) / np.float32(8)               # quantised like DAC values: plenty of duplicates


def test_reference_fixtures(B, engine):
    sig = np.load(os.path.join(HERE, "golden", "signal_fixtures.npz"))
    unit = np.float32(1373.41) / np.float32(8192.0)
    raw = ((sig["raw"].astype(np.float32) + np.float32(16.0)) * unit).astype(np.float32)     # test_flappie_signal.c:75-78
    p = B.Prepared(engine, [raw], trim_start=200, trim_end=10, varseg_chunk=100, varseg_thresh=0.0)
    s, e = p.range(0)
    assert (s, e) == (200, (raw.size // 100) * 100 - 10)                                    # :84-92
    assert np.abs(raw[s:e] - sig["trimmed"]).max() <= 1e-4                                  # trimmed_signal.crp
    got = p.signal(0)
    assert np.abs(got - sig["normalised"]).max() <= 1e-5                                    # normalised_signal.crp
    ref = oracle_prep(raw)
    assert (s, e) == ref[:2] and np.array_equal(got, ref[2])
    med, mad = p.stats(0)
    assert med == ffo.lib().fo_medianf(_f(raw[s:e].copy()), e - s)
    p.close()


def test_ragged_batch_bit_exact_against_oracle(B, engine):
    rng = np.random.default_rng(42)
    raws = [synth_raw(rng, 4000, lead=700, tail=300), synth_raw(rng, 4137, lead=350), synth_raw(rng, 2600),
            synth_raw(rng, 100000, lead=5000, tail=2500), synth_raw(rng, 399), synth_raw(rng, 40000, lead=1234),
            rng.standard_normal(5000).astype(np.float32)]
    for perc, chunk, trims in ((0.0, 100, (200, 10)), (0.5, 100, (0, 0)), (0.9, 50, (150, 20)), (0.25, 333, (10, 10)), (1.0, 100, (0, 0))):
        p = B.Prepared(engine, raws, trim_start=trims[0], trim_end=trims[1], varseg_chunk=chunk, varseg_thresh=perc)
        for i, raw in enumerate(raws):
            ref = oracle_prep(raw, trims[0], trims[1], chunk, perc)
            s, e = p.range(i)
            if ref is None:
                assert s >= e, (i, perc, chunk)
                continue
            assert (s, e) == ref[:2], (i, perc, chunk)
            assert np.array_equal(p.signal(i), ref[2]), (i, perc, chunk)
        p.close()


def test_rejected_and_degenerate_reads(B, engine):
    rng = np.random.default_rng(1)
    flat = np.full(1000, 80.0, dtype=np.float32)                     # every chunk MAD is 0: nothing exceeds the threshold
    short = synth_raw(rng, 150)                                      # one chunk, then trimmed away by 200:10
    ok = synth_raw(rng, 3000)
    p = B.Prepared(engine, [flat, short, ok])
    for i, raw in enumerate((flat, short)):
        assert oracle_prep(raw) is None
        s, e = p.range(i)
        assert s >= e
        with pytest.raises(B.FFHipError):
            p.signal(i)
    ref = oracle_prep(ok)
    assert p.range(2) == ref[:2] and np.array_equal(p.signal(2), ref[2])
    p.close()
    with pytest.raises(B.FFHipError):
        B.Prepared(engine, [ok], varseg_chunk=1)                     # assert(chunk_size > 1), flappie_common.c:48
    with pytest.raises(B.FFHipError):
        B.Prepared(engine, [ok], varseg_thresh=1.5)


def test_delta_mode_and_array_entry_points(B, engine):
    L = B.lib()
    rng = np.random.default_rng(9)
    raw = synth_raw(rng, 3000, lead=400)
    p = B.Prepared(engine, [raw], mode=B.PREP_DELTA, delta=2.5)
    s, e = p.range(0)
    y = raw[s:e]
    want = np.concatenate([y[1:] - y[:-1], [np.float32(0)]]).astype(np.float32) / np.float32(2.5)      # flappie.c:261-262
    assert np.array_equal(p.signal(0), want)
    p.close()
    # quantilef / madf / medmad_normalise_array on single arrays
    for n in (1, 2, 5, 100, 4001, 65536):
        x = synth_raw(rng, n) if n > 2 else rng.standard_normal(n).astype(np.float32)
        q = np.array([0.0, 0.25, 0.5, 0.9, 1.0], dtype=np.float32)
        want = q.copy()
        ffo.lib().fo_quantilef(_f(x), n, _f(want), q.size)
        got = q.copy()
        assert L.ffhip_quantiles(engine.h, _f(x), n, _f(got), q.size) == 0
        assert np.array_equal(got, want), n
        mad = C.c_float(0)
        assert L.ffhip_mad(engine.h, _f(x), n, None, C.byref(mad)) == 0
        assert mad.value == ffo.lib().fo_madf(_f(x), n, None)
        med = C.c_float(1.25)
        assert L.ffhip_mad(engine.h, _f(x), n, C.byref(med), C.byref(mad)) == 0
        assert mad.value == ffo.lib().fo_madf(_f(x), n, C.byref(med))
        a, b = x.copy(), x.copy()
        assert L.ffhip_medmad_normalise(engine.h, _f(a), n, None, None) == 0
        ffo.lib().fo_medmad_normalise_array(_f(b), n)
        assert np.array_equal(a, b, equal_nan=True), n


def test_prepared_reads_feed_batches_device_to_device(B, engine):
    """raw reads -> GPU prep -> length buckets -> batches, no host round trip of the signal; the calls equal
    those of the host-fed path on the oracle's prepared signal."""
    from flappie_amd import model as M
    rng = np.random.default_rng(5)
    mdl = M.synthetic_model(M.NET_LSTM5, 48, seed=3)
    dm = B.DeviceModel(engine, mdl)
    raws = [synth_raw(rng, n, lead=l) for n, l in ((4000, 600), (4000, 600), (3000, 0), (4000, 600), (3000, 0))]
    p = B.Prepared(engine, raws)
    lens = {}
    for i in range(len(raws)):
        s, e = p.range(i)
        lens.setdefault(e - s, []).append(i)
    assert len(lens) >= 2
    om = ffo.OracleModel(mdl)
    for n, idx in lens.items():
        b = B.Batch(dm, len(idx), n)
        b.set_prepared(p, idx)
        b.run(); b.finish()
        for k, i in enumerate(idx):
            ref = om.basecall(oracle_prep(raws[i])[2])
            assert b.basecall(k) == ref["basecall"] and b.quality(k) == ref["quality"]
        other = [j for j in range(len(raws)) if j not in idx][:1] * len(idx)
        if p.range(other[0])[1] - p.range(other[0])[0] > n:
            with pytest.raises(B.FFHipError):
                b.set_prepared(p, other)                                   # longer than the batch's capacity
        else:
            b.set_prepared(p, other)                                       # shorter: a ragged batch (tests/test_ragged_gpu.py)
        b.close()
    p.close()
    dm.close()


def test_begin_and_finish_equal_create(B, engine):
    """ffhip_prep_begin + ffhip_prep_finish (the two halves a pipeline uses to prepare a chunk ahead) give what ffhip_prep_create gives: ranges, statistics, signals bit for bit;
    a second finish is a no-op, and a preparation begun and never finished is waited for by its destruction"""
    rng = np.random.default_rng(21)
    raws = [(500 + 60 * rng.standard_normal(n)).astype(np.float32) for n in (1500, 4000, 4013, 777, 20000, 2600)]
    for r in raws:
        r[:300] = (520 + 4 * rng.standard_normal(300)).astype(np.float32)
    a = B.Prepared(engine, raws)
    b = B.Prepared(engine, raws, begin_only=True)
    b.finish()
    b.finish()
    for i in range(len(raws)):
        assert a.range(i) == b.range(i) and a.stats(i) == b.stats(i)
        assert np.array_equal(a.signal(i), b.signal(i))
    c = B.Prepared(engine, raws, begin_only=True)
    c.close()
    a.close()
    b.close()

