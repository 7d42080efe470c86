"""A/B of the CRF head: f32-MFMA head on the last layer's fp32 copy (FFHIP_DEBUG=no_split_head) against k_head_split -- scores vs oracle, time."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from flappie_amd import binding as B, model as M
    from oracle import ffo
    eng = B.Engine(0)
    out = {}
    for kind, H, nread, T in ((0, 384, 256, 4000), (1, 256, 1024, 2000), (0, 512, 48, 3000), (2, 384, 256, 4000), (0, 128, 33, 1234)):
        mdl = M.synthetic_model(kind, H, seed=1)
        dm = B.DeviceModel(eng, mdl)
        rng = np.random.default_rng(H + nread)
        sig = rng.standard_normal((nread, T)).astype(np.float32)
        b = B.Batch(dm, nread, T)
        b.set_signals(sig)
        b.run(); b.finish()
        om = ffo.OracleModel(mdl)
        worst = 0.0
        same = True
        for r in (0, nread // 2, nread - 1):
            if kind == 2:
                ref = om.runlength_call(sig[r, :1500]) if False else None
                continue
            ref = om.basecall(sig[r])
            worst = max(worst, float(np.abs(b.transitions(r) - ref["trans"]).max()))
            same = same and b.basecall(r) == ref["basecall"]
        tr0 = b.transitions(0).copy()
        eng.set_profiling(True)
        for _ in range(3):
            b.run(); b.finish()
        p = b.profile()
        eng.set_profiling(False)
        out["%d/%d/%d/%d" % (kind, H, nread, T)] = (worst, same, round(p["head_crf"]["ms"], 4), round(p["recurrent"]["ms"], 3), float(np.abs(tr0).sum()))
        b.close(); dm.close()
    print(json.dumps(out))
else:
    res = {}
    for tag, env in (("f32 head", {"FFHIP_DEBUG": "no_split_head"}), ("split head", {})):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        if r.returncode != 0:
            print(tag, "failed:", r.stderr[-3000:]); sys.exit(1)
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    for k in res["f32 head"]:
        a, c = res["f32 head"][k], res["split head"][k]
        print("%-20s |dtrans| vs oracle %.2e -> %.2e, bases equal %s -> %s; head+crf group %.4f -> %.4f ms; 5 layers %.3f -> %.3f ms; checksum %.6g / %.6g"
              % (k, a[0], c[0], a[1], c[1], a[2], c[2], a[3], c[3], a[4], c[4]))
