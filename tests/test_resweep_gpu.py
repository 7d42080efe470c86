"""The layer kernels' RE-SWEEP path, taken on purpose.

A sweep of h(t-1) that meets a sentinel (a producer's stores became visible line by line) is done again; that happens ~10 times a layer launch (0.001 % of the h waves' steps, tools/dev/fallback_count.py), and
for two rounds the path was wrong: the re-sweep's `buffer_load ... lds` pieces went out while the failed pass's last partial-sum `ds_write`s into the same landing
zone -- the SECOND tile's -- were still queued, the two LDS write paths are not ordered, and a stale partial could replace an operand piece: one read tile wrong from
that step on, about one launch in 10^4 (round 5: found with tools/dev/pack_repeat.py, 2 of 10 000 launches of k_grumod_pack; 199 of 200 with the path forced; profiles/r05_pack_repeat.txt).
tools/test_hooks/libffhip_resweep.so is the release library with -DFFHIP_FORCE_RETRY=1 (every member re-sweeps once at every 32nd step): each dense / packed / paired
form must give, bit for bit, what the release library gives.

Round 6 adds the guard for the CLASS rather than the instance: tools/test_hooks/libffhip_skew.so, the release library with -DFFHIP_FORCE_SKEW=1 -- at
every phase boundary of the layer kernels' step (top, behind the poll, before / behind barrier 1, behind the gate phase, behind barrier 2) one wave in thirteen, rotating
with step, site, wave and group member, sits out a third of a step; in k_conv_split_ws (the other kernel that mixes LDS-DMA writes, DS reads and an LDS-only barrier) one
wave in seven at four sites of the tile loop.  The same shapes, the same bar.  (k_head_split exchanges values between lanes of ONE wave only -- `__shfl_xor` -- and has no
LDS: there is nothing a delayed wave could expose there.)"""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOK = os.environ.get("FFHIP_TEST_RESWEEP_LIB") or os.path.join(ROOT, "tools", "test_hooks", "libffhip_resweep.so")      # (the variable: to show that the test fails on a build without the fix)
SKEW = os.path.join(ROOT, "tools", "test_hooks", "libffhip_skew.so")

CHILD = r"""
import os, pickle, sys
import numpy as np
sys.path.insert(0, %(root)r)
from flappie_amd import binding as B
from flappie_amd import model as M
kind, hidden, nread, T, pair, reps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
packed = len(sys.argv) > 8 and sys.argv[8] == "1"      # the same rows as a PACKED batch (several reads to a row: the LIVE instantiations of the layer kernels)
eng = B.Engine(0)
dm = B.DeviceModel(eng, M.synthetic_model(kind, hidden, seed=5 + kind))
rng = np.random.default_rng(nread + hidden)
lens = rng.integers(T // 5, T + 1, size=nread); lens[:3] = (T, T // 5, T // 4)
sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
if packed:
    lens = rng.integers(T // 8, T // 2, size=3 * nread); lens[:3] = (T - 8, T // 8, T // 4)
    sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
res = []
for rep in range(reps):
    bs = [B.Batch(dm, nread, T, max_reads=len(sigs) if packed else 0) for _ in range(2 if pair else 1)]
    for b in bs:
        if packed:
            slot, off = b.pack_plan([x.size for x in sigs])
            keep = [i for i in range(len(sigs)) if slot[i] >= 0]
            assert len(keep) > 2 * nread
            b.set_signals_packed([sigs[i] for i in keep], [slot[i] for i in keep], [off[i] for i in keep])
        else:
            b.set_signals_ragged(sigs)
    if pair:
        bs[0].run_pair(bs[1])
    else:
        bs[0].run()
    for b in bs:
        b.finish()
    res.append([[(b.transitions(r).tobytes(), b.basecall(r), b.quality(r)) for r in range(b.nreads() if packed else nread)] for b in bs])
    for b in bs:
        b.close()
pickle.dump(res, open(out, "wb"))
"""


def _run(lib, kind, hidden, nread, T, pair, reps, out, packed=False):
    env = dict(os.environ)
    env.pop("FFHIP_DEBUG", None)
    if lib:
        env["FFHIP_BINDING_LIBRARY"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(kind), str(hidden), str(nread), str(T), str(int(pair)), str(reps), out, str(int(packed))], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return pickle.load(open(out, "rb"))


@pytest.mark.parametrize("lib", [HOOK, SKEW])
def test_the_resweep_library_is_built_and_is_another_build_of_the_same_abi(lib):
    assert os.path.exists(lib), "%s is missing: __graft_entry__.build() (make hooks) builds it" % os.path.relpath(lib, ROOT)
    import ctypes
    L = ctypes.CDLL(lib)
    for sym in ("ffhip_batch_run", "ffhip_batch_run_pair", "ffhip_engine_create"):
        assert hasattr(L, sym)


# (cell kind, hidden, reads, samples, in pairs): k_grumod_pack + a 16-read rest, k_lstm_pack + rest, the dense H = 256 forms (768 reads), H = 384 in pairs (the headline's
# kernel) and alone, H = 512
@pytest.mark.gpu
@pytest.mark.parametrize("kind,hidden,nread,T,pair", [(1, 256, 1040, 1000, False), (0, 256, 1040, 1000, False), (1, 256, 768, 1000, False), (0, 384, 256, 1500, True),
                                                      (0, 384, 256, 1500, False), (0, 512, 256, 1000, False)])
def test_a_resweep_changes_nothing(tmp_path, kind, hidden, nread, T, pair):
    ref = _run(None, kind, hidden, nread, T, pair, 1, str(tmp_path / "ref.pkl"))[0]
    got = _run(HOOK, kind, hidden, nread, T, pair, 12, str(tmp_path / "got.pkl"))
    for rep, bs in enumerate(got):
        for k, b in enumerate(bs):
            bad = [r for r in range(nread) if b[r] != ref[k][r]]
            assert not bad, "launch %d, batch %d: %d reads differ (tiles %s)" % (rep, k, len(bad), sorted(set(r // 16 for r in bad))[:8])



# the skewed build: the same forms + the H = 128 one-tile form; every LSTM case runs k_conv_split_ws (its last convolution) skewed as well
@pytest.mark.gpu
@pytest.mark.parametrize("kind,hidden,nread,T,pair", [(1, 256, 1040, 1000, False), (0, 256, 1040, 1000, False), (0, 384, 256, 1500, True), (0, 384, 256, 1500, False),
                                                      (0, 512, 256, 1000, False), (0, 128, 96, 1000, False)])
def test_a_late_wave_changes_nothing(tmp_path, kind, hidden, nread, T, pair):
    ref = _run(None, kind, hidden, nread, T, pair, 1, str(tmp_path / "ref.pkl"))[0]
    got = _run(SKEW, kind, hidden, nread, T, pair, 4, str(tmp_path / "got.pkl"))
    for rep, bs in enumerate(got):
        for k, b in enumerate(bs):
            bad = [r for r in range(nread) if b[r] != ref[k][r]]
            assert not bad, "launch %d, batch %d: %d reads differ (tiles %s)" % (rep, k, len(bad), sorted(set(r // 16 for r in bad))[:8])


# ... and the PACKED instantiations of the same forms (round 6: several reads to a row, the live mask; kernels of their own): the paired launch, the two packed H = 256 forms in
# full 1024-row launches, the one-tile form -- re-swept on purpose and with late waves
@pytest.mark.gpu
@pytest.mark.parametrize("lib,reps", [(HOOK, 6), (SKEW, 3)])
@pytest.mark.parametrize("kind,hidden,nread,T,pair", [(0, 384, 256, 1500, True), (0, 256, 1024, 1000, False), (1, 256, 1024, 1000, False), (0, 128, 96, 1000, False)])
def test_packed_batches_under_a_resweep_and_with_late_waves(tmp_path, lib, reps, kind, hidden, nread, T, pair):
    ref = _run(None, kind, hidden, nread, T, pair, 1, str(tmp_path / "ref.pkl"), packed=True)[0]
    got = _run(lib, kind, hidden, nread, T, pair, reps, str(tmp_path / "got.pkl"), packed=True)
    for rep, bs in enumerate(got):
        for k, b in enumerate(bs):
            assert len(b) == len(ref[k])
            bad = [r for r in range(len(b)) if b[r] != ref[k][r]]
            assert not bad, "launch %d, batch %d: %d reads differ" % (rep, k, len(bad))
