"""The same launch, many times: every one must give the first one's bits.

Round 5's fault (a re-sweep of h(t-1) racing a queued ds_write: one read tile wrong from some step on, about one launch in 10^4) was FOUND by launching one ragged packed
batch thousands of times (tools/dev/pack_repeat.py) after 184 green tests had missed it twice; tests/test_resweep_gpu.py pins that instance with a forced-path build.  This
is the stress itself in short form, so that the CLASS -- a rare interleaving inside a persistent layer launch -- has a standing guard: REPS launches of
  * a ragged 1040-read batch through k_grumod_pack and through k_lstm_pack (+ their 16-read rests on the one-tile form),
  * two ragged 256-read batches as one paired launch per layer, k_lstm_split_pair<0,3,2,true> -- the kernel bench.py times,
every read's called bases, qualities and Viterbi score (a digest of the read's whole evaluation: a tile that goes wrong at any step moves it) against the ONE-TILE launches
of the same reads (FFHIP_DEBUG=no_dense,no_pair), whose landing zones and partial sums are apart by construction.  300 launches see a 1-in-100 fault with probability 0.95
and a 1-in-10^4 one hardly ever: the long form stays a tool (profiles/r05_pack_repeat.txt: 20 000 launches of the shipping library, none deviates)."""
import os

import numpy as np
import pytest

from flappie_amd import binding as B
from flappie_amd import model as M

REPS = int(os.environ.get("FFHIP_TEST_REPEAT_LAUNCHES", "300"))


def _results(b, nread):
    return [(b.basecall(r), b.quality(r), b.score(r)) for r in range(nread)]


@pytest.fixture(scope="module")
def eng():
    e = B.Engine(0)
    yield e
    e.close()


def _reads(nread, T, seed):
    rng = np.random.default_rng(seed)
    lens = rng.integers(T // 5, T + 1, size=nread)
    lens[:3] = (T, T // 5, T // 5 + 37)
    return [rng.standard_normal(int(n)).astype(np.float32) for n in lens]


def _with_debug(tokens, fn):
    old = os.environ.get("FFHIP_DEBUG")
    os.environ["FFHIP_DEBUG"] = tokens
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop("FFHIP_DEBUG")
        else:
            os.environ["FFHIP_DEBUG"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [M.NET_GRUMOD5, M.NET_LSTM5])
def test_packed_launches_repeat_bit_for_bit(eng, kind):
    nread, T = 1040, 1000
    dm = B.DeviceModel(eng, M.synthetic_model(kind, 256, seed=5 + kind))
    sigs = _reads(nread, T, nread)
    b = B.Batch(dm, nread, T)

    def once():
        b.set_signals_ragged(sigs)
        b.run()
        b.finish()
        return _results(b, nread)

    ref = _with_debug("no_dense", once)           # one tile per group: partial sums and landing zones apart
    deviating = []
    for rep in range(REPS):
        got = once()
        assert b.rnn_path() == 3
        bad = [r for r in range(nread) if got[r] != ref[r]]
        if bad:
            deviating.append((rep, len(bad), sorted(set(r // 16 for r in bad))[:6]))
    b.close()
    dm.close()
    assert not deviating, "%d of %d launches deviate from the one-tile launches: (launch, reads, tiles) %s" % (len(deviating), REPS, deviating[:5])


@pytest.mark.gpu
def test_paired_headline_launches_repeat_bit_for_bit(eng):
    nread, T = 256, 1500
    dm = B.DeviceModel(eng, M.synthetic_model(M.NET_LSTM5, 384, seed=1))
    sigs = [_reads(nread, T, 77), _reads(nread, T, 78)]
    bs = [B.Batch(dm, nread, T) for _ in range(2)]

    def alone():
        out = []
        for k in (0, 1):
            bs[k].set_signals_ragged(sigs[k])
            bs[k].run()
            bs[k].finish()
            out.append(_results(bs[k], nread))
        return out

    ref = _with_debug("no_dense,no_pair,split_dense=0", alone)
    deviating = []
    for rep in range(REPS):
        for k in (0, 1):
            bs[k].set_signals_ragged(sigs[k])
        bs[0].run_pair(bs[1])
        for k in (0, 1):
            bs[k].finish()
            assert bs[k].paired()
            got = _results(bs[k], nread)
            bad = [r for r in range(nread) if got[r] != ref[k][r]]
            if bad:
                deviating.append((rep, k, len(bad), sorted(set(r // 16 for r in bad))[:6]))
    for b in bs:
        b.close()
    dm.close()
    assert not deviating, "%d of %d paired launches deviate from the one-tile launches: (launch, batch, reads, tiles) %s" % (len(deviating), REPS, deviating[:5])


@pytest.mark.gpu
def test_paired_launches_of_packed_batches_repeat_bit_for_bit(eng):
    """the same stress on the PACKED instantiation of the headline's kernel (round 6: several reads to a row, the live mask): two packed 256-row batches as one paired
    launch per layer, REPS times, every read against the ONE-TILE launches of the same packed rows"""
    rows, T = 256, 1500
    dm = B.DeviceModel(eng, M.synthetic_model(M.NET_LSTM5, 384, seed=1))
    rng = np.random.default_rng(81)
    bs, sets = [], []
    for k in (0, 1):
        lens = [int(x) for x in np.clip(np.exp(np.log(300) + 0.8 * rng.standard_normal(800)), 25, T - 50)]
        sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
        b = B.Batch(dm, rows, T, max_reads=len(lens))
        slot, off = b.pack_plan(lens)
        keep = [i for i in range(len(lens)) if slot[i] >= 0]
        bs.append(b)
        sets.append(([sigs[i] for i in keep], [slot[i] for i in keep], [off[i] for i in keep]))

    def alone():
        out = []
        for k in (0, 1):
            bs[k].set_signals_packed(*sets[k])
            bs[k].run()
            bs[k].finish()
            out.append(_results(bs[k], bs[k].nreads()))
        return out

    ref = _with_debug("no_dense,no_pair,split_dense=0", alone)
    deviating = []
    for rep in range(REPS):
        for k in (0, 1):
            bs[k].set_signals_packed(*sets[k])
        bs[0].run_pair(bs[1])
        for k in (0, 1):
            bs[k].finish()
            assert bs[k].paired()
            got = _results(bs[k], bs[k].nreads())
            bad = [r for r in range(len(got)) if got[r] != ref[k][r]]
            if bad:
                deviating.append((rep, k, len(bad)))
    for b in bs:
        b.close()
    dm.close()
    assert not deviating, "%d of %d paired launches of packed batches deviate from the one-tile launches: (launch, batch, reads) %s" % (len(deviating), REPS, deviating[:5])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [M.NET_GRUMOD5, M.NET_LSTM5])
def test_full_packed_h256_launches_of_packed_batches_repeat_bit_for_bit(eng, kind):
    """... and on the PACKED-BATCH instantiations of the two H = 256 forms (k_grumod_pack<true, 2>, k_lstm_pack<true, 2>): a full 1024-row launch whose rows hold
    several reads each, REPS times, against the one-tile launches of the same rows"""
    rows, T = 1024, 1000
    dm = B.DeviceModel(eng, M.synthetic_model(kind, 256, seed=5 + kind))
    rng = np.random.default_rng(91 + kind)
    lens = [int(x) for x in np.clip(np.exp(np.log(220) + 0.7 * rng.standard_normal(3200)), 25, T - 50)]
    sigs = [rng.standard_normal(n).astype(np.float32) for n in lens]
    b = B.Batch(dm, rows, T, max_reads=len(lens))
    slot, off = b.pack_plan(lens)
    keep = [i for i in range(len(lens)) if slot[i] >= 0]
    args_ = ([sigs[i] for i in keep], [slot[i] for i in keep], [off[i] for i in keep])
    assert len(keep) > 2 * rows and len(set(slot[i] for i in keep)) > 900        # (nearly) every row in use; the batch has 64 read tiles either way: the launch is a full one of the packed form

    def once():
        b.set_signals_packed(*args_)
        b.run()
        b.finish()
        return _results(b, b.nreads())

    ref = _with_debug("no_dense,no_pack", once)
    deviating = []
    for rep in range(REPS):
        got = once()
        assert b.rnn_path() == 3
        bad = [r for r in range(len(got)) if got[r] != ref[r]]
        if bad:
            deviating.append((rep, len(bad)))
    b.close()
    dm.close()
    assert not deviating, "%d of %d launches deviate from the one-tile launches: (launch, reads) %s" % (len(deviating), REPS, deviating[:5])
