#!/usr/bin/env python3
"""Who is closer to exact arithmetic?  The recurrent stack + head of the LSTM5 network evaluated in float64 numpy (from the
oracle's own float32 convolution trunk, which is not recurrent and is common to all), against (a) the oracle -- float32 with the
reference's sequential dot products --, (b) the GPU split-bf16 path, (c) the GPU f32-MFMA path.  Differences between (a) and
(b)/(c) beyond 1e-4 on random models (tools/dev/diff_fuzz.py) are put in perspective by their distances from the exact result.
usage: fp64_truth.py [nread] [nsample] [H ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import binding as B, model as M  # noqa: E402
from oracle import ffo  # noqa: E402


def trunk(om, sig):
    """the oracle's convolution trunk (float32): [T, H]"""
    L = ffo.lib()
    L.fo_features_from_raw.restype = C.POINTER(ffo.FoMat)
    L.fo_features_from_raw.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t]
    sig = np.ascontiguousarray(sig, dtype=np.float32)
    x = L.fo_features_from_raw(sig.ctypes.data_as(C.POINTER(C.c_float)), 0, sig.size)
    for i in range(om.c.nconv):
        y = L.fo_convolution(x, om.c.conv_W[i], om.c.conv_b[i], om.c.conv_stride[i])
        L.fo_swish_inplace(y)
        L.fo_free_mat(x)
        x = y
    return ffo.take(x)


def sigma(v):
    return 1.0 / (1.0 + np.exp(-v))


def exact_trans(mdl, x, temperature=1.0):
    """float64: 5 LSTM layers (B, F, B, F, B; networks.c:539-586), head, global normalisation.  x [nread, T, H]"""
    x = x.astype(np.float64)
    nread, T, H = x.shape
    for l, r in enumerate(mdl.rnns):
        iW, sW, b = r.iW.dense().astype(np.float64), r.sW.dense().astype(np.float64), r.b.dense().astype(np.float64).reshape(-1)
        xa = x @ iW.T + b
        out = np.zeros((nread, T, H))
        h, c = np.zeros((nread, H)), np.zeros((nread, H))
        for i in range(T):
            t = T - 1 - i if l % 2 == 0 else i
            g = xa[:, t] + h @ sW.T
            c = sigma(g[:, H:2 * H]) * c + sigma(g[:, :H]) * np.tanh(g[:, 2 * H:3 * H])
            h = sigma(g[:, 3 * H:]) * np.tanh(c)
            out[:, t] = h
        x = out
    W, b = mdl.FF_W.dense().astype(np.float64), mdl.FF_b.dense().astype(np.float64).reshape(-1)
    s = 5.0 * np.tanh(x @ W.T + b) / temperature                      # [nread, T, P]
    nbase = mdl.nbase
    ns = 2 * nbase
    prev = np.zeros((nread, ns))
    for t in range(T):                                                # layers.c:1035-1079
        col = s[:, t]
        stay = col[:, ns * nbase:]
        cur = np.empty_like(prev)
        cur[:, nbase:] = np.logaddexp(prev[:, nbase:] + stay[:, nbase:], prev[:, :nbase] + stay[:, :nbase])
        rows = col[:, :ns * nbase].reshape(nread, nbase, ns) + prev[:, None, :]
        cur[:, :nbase] = np.logaddexp.reduce(rows, axis=2)
        prev = cur
    logz = np.logaddexp.reduce(prev, axis=1)
    return s - (logz / T)[:, None, None]


def main():
    nread = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
    hs = [int(v) for v in sys.argv[3:]] or [256, 384]
    eng = B.Engine(0)
    print("%-22s %12s %12s %12s %14s %14s" % ("model", "oracle-exact", "split-exact", "f32-exact", "split-oracle", "f32-oracle"))
    for H in hs:
        for seed in range(100, 106):
            mdl = M.synthetic_model(M.NET_LSTM5, H, seed=seed)
            om = ffo.OracleModel(mdl)
            dm = B.DeviceModel(eng, mdl)
            rng = np.random.default_rng(seed)
            sig = rng.standard_normal((nread, nsample)).astype(np.float32)
            x = np.stack([trunk(om, sig[r]) for r in range(nread)])
            exact = exact_trans(mdl, x)
            orc = np.stack([om.transitions(sig[r]) for r in range(nread)])
            gpu = {}
            for name, flags in (("split", 0), ("f32", B.RUN_F32_RNN)):
                b = B.Batch(dm, nread, nsample)
                b.set_signals(sig)
                b.run(1.0, flags)
                b.finish()
                gpu[name] = np.stack([b.transitions(r) for r in range(nread)])
                b.close()
            d = lambda a, c: float(np.abs(a.astype(np.float64) - c).max())
            print("LSTM5 H %3d seed %3d   %12.2e %12.2e %12.2e %14.2e %14.2e" % (H, seed, d(orc, exact), d(gpu["split"], exact), d(gpu["f32"], exact),
                                                                                 d(gpu["split"], orc), d(gpu["f32"], orc)), flush=True)
            dm.close()


if __name__ == "__main__":
    main()
