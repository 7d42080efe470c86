#!/usr/bin/env python3
"""How often does a called base string differ between two float evaluations of the same network, by decode mode?
Reads of bench.py's headline shape (LSTM5 H = 384, model seed 1) through ffhip_batch_run_pair and through the oracle twice -- dot mode 0 (float sums in index
order) and dot mode 3 (the reference's cblas_sgemv / cblas_sgemm shapes in a real OpenBLAS) -- at temperature 0.7 / 1.0 / 1.3, decoded from the POSTERIOR
(the reference's default, flappie.c:262-268) and from the SCORES (--viterbi, flappie.c:245-316 viterbi_only; decode.c:119-204).  Reported per mode:
reads and bases apart per million bases for engine <-> oracle, engine <-> oracle+blas, oracle <-> oracle+blas.

  tools/parity_modes.py oracle OUT.npz [nread=2048] [tmax=2500]     CPU, process pool (6 network evaluations a read; ~50 min for 2048 reads on 8 cores)
  tools/parity_modes.py gpu IN.npz                                  GPU box
Read r: default_rng(7000 + r), length uniform in [3 tmax / 5, tmax]."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

KIND, H, SEED = M.NET_LSTM5, 384, 1
TEMPS = (0.7, 1.0, 1.3)
_om = None


def signal(r, tmax):
    rng = np.random.default_rng(7000 + r)
    return rng.standard_normal(int(rng.integers(tmax * 3 // 5, tmax + 1))).astype(np.float32)


def bases_of_path(path, nbase=4):
    """change_positions (decode.c:66-79) + the base assembly of flappie.c:284-292"""
    p = np.asarray(path)
    nblock = p.size - 1
    pos = np.nonzero(p[1:nblock] != p[:nblock - 1])[0] + 1
    return "".join("ACGTZ"[int(p[k]) % nbase] for k in pos)


def _init():
    global _om
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle import ffo
    _om = ffo.OracleModel(M.synthetic_model(KIND, H, seed=SEED))


def _call(job):
    r, tmax = job
    import ctypes as C
    from oracle import ffo
    x = signal(r, tmax)
    out = {}
    for mode in (0, 3):
        got, _ = ffo.use_dot_mode(mode)
        assert got == mode, "no LP64 OpenBLAS on this host"
        for T in TEMPS:
            o = _om.basecall(x, temperature=T)
            out[(mode, T, "post")] = o["basecall"]
            nblock = o["nblock"]
            path = np.zeros(nblock + 2, dtype=np.int32)
            qpath = np.zeros(nblock + 2, dtype=np.float32)
            hm = ffo.HostMat.from_dense(o["trans"])
            ffo.lib().fo_decode_viterbi(hm.ptr, 0, path.ctypes.data_as(C.POINTER(C.c_int)), qpath.ctypes.data_as(C.POINTER(C.c_float)))
            out[(mode, T, "vit")] = bases_of_path(path[:nblock + 1])
    ffo.use_dot_mode(0)
    return r, out


def cmd_oracle(out, nread, tmax):
    t0 = time.time()
    store = {"nread": np.int64(nread), "tmax": np.int64(tmax)}
    with mp.Pool(len(os.sched_getaffinity(0)), initializer=_init) as pool:
        for k, (r, o) in enumerate(pool.imap_unordered(_call, [(r, tmax) for r in range(nread)], chunksize=4)):
            for (mode, T, dec), s in o.items():
                store["r%d_m%d_T%.1f_%s" % (r, mode, T, dec)] = np.asarray(s)
            if k % 128 == 127:
                print("%d reads, %.0f s" % (k + 1, time.time() - t0), flush=True)
    np.savez_compressed(out, **store)
    print("wrote %s: %d reads, %.0f s" % (out, nread, time.time() - t0))


def apart(a, b):
    if a == b:
        return 0
    import difflib
    sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
    return max(len(a), len(b)) - sum(m.size for m in sm.get_matching_blocks())


def cmd_gpu(inp):
    z = np.load(inp, allow_pickle=False)
    nread, tmax = int(z["nread"]), int(z["tmax"])
    from flappie_amd import binding as B
    eng = B.Engine(0)
    dm = B.DeviceModel(eng, M.synthetic_model(KIND, H, seed=SEED))
    assert nread % 512 == 0
    sigs = [signal(r, tmax) for r in range(nread)]
    bs = [B.Batch(dm, 256, tmax) for _ in range(2)]
    print("near-tie rate by decode mode: %d reads of %d..%d samples, LSTM5 H = 384 (bench.py's model), engine through ffhip_batch_run_pair" % (nread, tmax * 3 // 5, tmax))
    for T in TEMPS:
        for dec, flags in (("post", 0), ("vit", B.RUN_VITERBI_ONLY)):
            eng_calls = [None] * nread
            for k in range(0, nread, 512):
                for j in (0, 1):
                    bs[j].set_signals_ragged(sigs[k + 256 * j:k + 256 * (j + 1)])
                bs[0].run_pair(bs[1], T, flags)
                for j in (0, 1):
                    bs[j].finish()
                    assert bs[j].rnn_path() == 3
                    for r in range(256):
                        eng_calls[k + 256 * j + r] = bs[j].basecall(r)
            nb = 0
            cnt = {"engine <-> oracle": [0, 0], "engine <-> oracle+blas": [0, 0], "oracle <-> oracle+blas": [0, 0]}
            both = 0
            for r in range(nread):
                o0, o3 = str(z["r%d_m0_T%.1f_%s" % (r, T, dec)]), str(z["r%d_m3_T%.1f_%s" % (r, T, dec)])
                nb += len(o0)
                d = (apart(eng_calls[r], o0), apart(eng_calls[r], o3), apart(o0, o3))
                for key, v in zip(cnt, d):
                    if v:
                        cnt[key][0] += 1
                        cnt[key][1] += v
                both += 1 if (d[0] and d[1]) else 0
            print("temperature %.1f, decode of the %s: %d bases" % (T, "posterior" if dec == "post" else "scores (--viterbi)", nb))
            for key, (nr, nbp) in cnt.items():
                print("    %-24s %d reads with another base string, %d bases apart = %.1f per million bases" % (key + ":", nr, nbp, 1e6 * nbp / max(1, nb)))
            print("    reads the engine calls differently from BOTH oracle evaluations: %d" % both)
    for b in bs:
        b.close()
    dm.close(); eng.close()


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "oracle":
        cmd_oracle(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2048, int(sys.argv[4]) if len(sys.argv) > 4 else 2500)
    elif len(sys.argv) >= 3 and sys.argv[1] == "gpu":
        cmd_gpu(sys.argv[2])
    else:
        sys.exit(__doc__)
