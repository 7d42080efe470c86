"""A/B of a convolution switch -- bit-identity of everything downstream and time.  Default: the last convolution, round-3 kernel (FFHIP_DEBUG=conv_ws=0)
against the weights-stationary one; `conv_ab.py FFHIP_DEBUG=conv_small_u` the thin front layers, round-3 loops (=0) against the unrolled ones."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from flappie_amd import binding as B, model as M
    eng = B.Engine(0)
    out = {}
    for kind, H, nread, T, ragged in ((0, 384, 256, 4000, False), (0, 384, 100, 1777, True), (0, 512, 48, 3001, False), (0, 256, 64, 999, True), (2, 384, 64, 2000, False), (0, 128, 33, 1234, True), (1, 256, 64, 1500, True), (1, 128, 40, 999, False), (1, 256, 1024, 4000, False)):
        mdl = M.synthetic_model(kind, H, seed=1)
        dm = B.DeviceModel(eng, mdl)
        rng = np.random.default_rng(H + nread)
        if ragged:
            lens = rng.integers(max(19, T // 3), T + 1, nread); lens[0] = T; lens[nread // 2] = 0
        else:
            lens = np.full(nread, T)
        sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
        b = B.Batch(dm, nread, T)
        b.set_signals_ragged(sigs)
        b.run(); b.finish()
        import hashlib
        h = hashlib.sha256()
        for r in range(nread):
            if lens[r]:
                h.update(b.transitions(r).tobytes())
        eng.set_profiling(True)
        for _ in range(3):
            b.run(); b.finish()
        p = b.profile()
        eng.set_profiling(False)
        out["%d/%d/%d/%d/%s" % (kind, H, nread, T, ragged)] = (h.hexdigest()[:16], round(p["conv"]["ms"], 4))
        b.close(); dm.close()
    print(json.dumps(out))
else:
    res = {}
    var = sys.argv[1] if len(sys.argv) > 1 else "conv_ws"      # a FFHIP_DEBUG token
    for ws in ((sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("0", "1")):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, FFHIP_DEBUG="%s=%s" % (var, ws)), capture_output=True, text=True)
        if r.returncode != 0:
            print("%s=%s failed:" % (var, ws), r.stderr[-2000:]); sys.exit(1)
        res[ws] = json.loads(r.stdout.strip().splitlines()[-1])
    v0, v1 = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("0", "1")
    for k in res[v0]:
        a, c = res[v0][k], res[v1][k]
        print("%-28s digest %s / %s %s   conv group %.4f -> %.4f ms" % (k, a[0], c[0], "identical" if a[0] == c[0] else "** DIFFERENT **", a[1], c[1]))
