"""The kernels bench.py times, compared DIRECTLY with the oracle on input-driven models.

  * k_lstm_split_pair<0,3,2,true>  -- H = 384, the layers of TWO 256-read batches as one launch (ffhip_batch_run_pair): bench.py's `c2`
  * k_lstm_split<0,4,2>            -- H = 512: bench.py's `c5`
  * k_lstm_pack / k_grumod_pack    -- H = 256, full 1024-read launches: `h256`, `c4` (tests/test_split_gpu.py::test_packed_kernels_against_oracle
                                      holds 28 reads of such a launch to the oracle; here every slot is live and 24 are checked)

The models are bench.py's (synthetic_model(kind, H, seed=1)); tests/test_cabi_and_model.py::test_synthetic_models_are_input_driven holds
them to >= 1 base per 12 samples and >= 100 distinct 5-mers per 4000-sample read.  Bounds: north_star's -- base string, quality string and
Viterbi path identical, transition scores within 5e-5 (half of north_star's 1e-4; measured worst 2.1e-5); every test prints its worst |dtrans| and what it compared."""
import numpy as np
import pytest

from conftest import oracle_calls
from flappie_amd import model as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from flappie_amd import binding
    return binding


def kmers(s, k=5):
    return len({s[i:i + k] for i in range(len(s) - k + 1)})


def compare(b, r, ref, stats):
    assert b.read_nblock(r) == ref["nblock"]
    d = float(np.abs(b.transitions(r) - ref["trans"]).max())
    stats["worst"] = max(stats["worst"], d)
    stats["reads"] += 1
    stats["bases"] += len(ref["basecall"])
    stats["kmers"] = min(stats["kmers"], kmers(ref["basecall"]))
    from conftest import note_parity
    note_parity(d, np.abs(b.posterior(r) - ref["post"]).max())
    assert d <= 5e-5, (r, d)
    assert b.basecall(r) == ref["basecall"], r
    assert b.quality(r) == ref["quality"], r
    assert np.array_equal(b.path(r)[0], ref["path"]), r
    assert np.abs(b.posterior(r) - ref["post"]).max() <= 1e-4          # (the scores' own deviation through two log-sum-exp recursions: tests/test_split_gpu.py)
    off = int((np.abs(b.trace(r) - ref["trace"]) > 0).sum())
    assert np.abs(b.trace(r) - ref["trace"]).max() <= 1
    stats["trace_off"] += off
    stats["trace_cells"] += ref["trace"].size


def new_stats():
    return dict(worst=0.0, reads=0, bases=0, kmers=10 ** 9, trace_off=0, trace_cells=0)


def report(what, st):
    print("%s: %d reads against the oracle, %d called bases (fewest distinct 5-mers in a read: %d), 0 base / quality / path mismatches, worst |dtrans| %.2e, "
          "trace off by one count in %d of %d cells" % (what, st["reads"], st["bases"], st["kmers"], st["worst"], st["trace_off"], st["trace_cells"]))


@pytest.mark.parametrize("ragged", [False, True])
def test_paired_launch_of_two_256_read_batches_against_the_oracle(B, engine, ragged):
    """bench.py's c2 step: two 256-read batches at H = 384 through ffhip_batch_run_pair -- k_lstm_split_pair<0,3,2,true> -- 12 reads of
    each batch (first and last slot, both tiles of a group, the tile boundary) against the oracle; uniform 2000-sample reads, and ragged
    reads of 600..2000 samples with empty slots"""
    mdl = M.synthetic_model(M.NET_LSTM5, 384, seed=1)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(3841 + ragged)
    T, nread = 2000, 256
    probe = [0, 1, 15, 16, 17, 31, 100, 127, 128, 200, 254, 255]
    batches, sigs = [], []
    for k in range(2):
        if ragged:
            lens = rng.integers(600, T + 1, nread)
            lens[rng.random(nread) < 0.1] = 0
            lens[probe] = rng.integers(600, T + 1, len(probe))
            lens[0] = T
            sg = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
        else:
            sg = list(rng.standard_normal((nread, T)).astype(np.float32))
        b = B.Batch(dm, nread, T)
        b.set_signals_ragged(sg)
        batches.append(b); sigs.append(sg)
    batches[0].run_pair(batches[1])
    st = new_stats()
    for b, sg in zip(batches, sigs):
        b.finish()
        assert b.paired() and b.rnn_path() == 3
        refs = oracle_calls(mdl, [sg[r] for r in probe])
        for r, ref in zip(probe, refs):
            compare(b, r, ref, st)
        b.close()
    dm.close()
    report("paired launch, H = 384, 2 x 256 reads%s" % (" (ragged)" if ragged else ""), st)
    assert st["reads"] == 24


def test_h512_layer_kernel_against_the_oracle(B, engine):
    """bench.py's c5 shape (r103: H = 512, k_lstm_split<0,4,2>): 48 reads of 1500 samples, 6 of them against the oracle"""
    mdl = M.synthetic_model(M.NET_LSTM5, 512, seed=1)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(512)
    sig = rng.standard_normal((48, 1500)).astype(np.float32)
    b = B.Batch(dm, 48, 1500)
    b.set_signals(sig)
    b.run(); b.finish()
    assert b.rnn_path() == 3
    probe = [0, 15, 16, 31, 40, 47]
    st = new_stats()
    for r, ref in zip(probe, oracle_calls(mdl, [sig[r] for r in probe])):
        compare(b, r, ref, st)
    b.close(); dm.close()
    report("H = 512 layer kernel", st)


@pytest.mark.parametrize("kind", [M.NET_LSTM5, M.NET_GRUMOD5])
def test_full_packed_launch_against_the_oracle(B, engine, kind):
    """bench.py's h256 / c4 step: a FULL 1024-read launch of the packed kernels (every slot live, uniform length), 24 reads against the oracle"""
    mdl = M.synthetic_model(kind, 256, seed=1)
    dm = B.DeviceModel(engine, mdl)
    assert dm.launch_reads == 1024
    T = 1500 if kind == M.NET_LSTM5 else 800
    sig = np.random.default_rng(2560 + kind).standard_normal((1024, T)).astype(np.float32)
    b = B.Batch(dm, 1024, T)
    b.set_signals(sig)
    b.run(); b.finish()
    assert b.rnn_path() == 3
    probe = [0, 1, 15, 16, 31, 32, 100, 255, 256, 300, 511, 512, 513, 600, 640, 700, 767, 768, 900, 990, 1000, 1008, 1022, 1023]
    st = new_stats()
    for r, ref in zip(probe, oracle_calls(mdl, [sig[r] for r in probe])):
        compare(b, r, ref, st)
    b.close(); dm.close()
    report("full packed launch, kind %d, H = 256" % kind, st)


def test_front_order_changes_the_schedule_not_the_results(B, engine):
    """bench.py's c2 pipeline -- two PAIRS in flight, the second pair's convolutions beside the first pair's head and decode
    (FFHIP_DEBUG=front_order=layers, the default; batch_run_impl) -- against round 3's order and against no order at all: every read's
    transition scores, calls and qualities byte for byte the same.  Three rounds through the same four batch objects, so that
    a batch's buffers are reused while its neighbours are still in flight."""
    import hashlib
    import os
    mdl = M.synthetic_model(M.NET_LSTM5, 384, seed=1)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(4004)
    T, nread = 1500, 256
    sigs = [rng.standard_normal((nread, T)).astype(np.float32) for _ in range(4)]
    batches = [B.Batch(dm, nread, T) for _ in range(4)]

    def digest(b):
        h = hashlib.sha256()
        for r in range(nread):
            h.update(b.transitions(r).tobytes()); h.update(b.basecall(r).encode()); h.update(b.quality(r).encode())
        return h.hexdigest()

    seen = {}
    old = os.environ.get("FFHIP_DEBUG")
    try:
        for order in ("layers", "batch", "none"):
            os.environ["FFHIP_DEBUG"] = "front_order=" + order
            got = []
            for rnd in range(3):
                for k in range(4):
                    batches[k].set_signals(sigs[(k + rnd) % 4])
                batches[0].run_pair(batches[1])
                batches[2].run_pair(batches[3])
                for b in batches:
                    b.finish()
                    assert b.paired()
                got.append([digest(batches[k]) for k in range(4)])
            # the same signal gives the same bytes in whichever batch object and round it ran
            for rnd in range(3):
                for k in range(4):
                    seen.setdefault((k + rnd) % 4, set()).add(got[rnd][k])
    finally:
        if old is None:
            os.environ.pop("FFHIP_DEBUG", None)
        else:
            os.environ["FFHIP_DEBUG"] = old
    for b in batches:
        b.close()
    dm.close()
    assert all(len(v) == 1 for v in seen.values()), {k: len(v) for k, v in seen.items()}
    assert len({next(iter(v)) for v in seen.values()}) == 4


def test_profile_of_a_pair_survives_role_swaps_and_destruction_order(B, engine):
    """a profiled pair brackets its layer launches once, on the first batch (two event records per launch, not six); the second batch reads
    them through a link that either batch's destruction and any later pairing take apart: same layer time from both, roles swapped, and the
    survivor's profile still answers after its mate is gone"""
    mdl = M.synthetic_model(M.NET_LSTM5, 384, seed=1)
    dm = B.DeviceModel(engine, mdl)
    rng = np.random.default_rng(77)
    sig = rng.standard_normal((256, 1000)).astype(np.float32)
    a, b, c = (B.Batch(dm, 256, 1000) for _ in range(3))
    for x in (a, b, c):
        x.set_signals(sig)
    engine.set_profiling(True)
    try:
        for first, second in ((a, b), (b, a), (c, a), (b, c)):
            first.run_pair(second)
            first.finish(); second.finish()
            p0, p1 = first.profile(), second.profile()
            assert first.paired() and second.paired()
            assert p0["recurrent"]["ms"] > 0.5 and p0["recurrent"]["ms"] == p1["recurrent"]["ms"]
        b.close()                      # the first batch of the last pair goes first: the second one's link is taken apart with it
        assert c.profile()["recurrent"]["ms"] >= 0.0
        a.run(); a.finish()            # and a batch that was half of a pair runs alone again
        assert not a.paired() and a.profile()["recurrent"]["ms"] > 0.5
    finally:
        engine.set_profiling(False)
    a.close(); c.close()
    dm.close()
