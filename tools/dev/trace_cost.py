"""what does --trace cost the binary on long reads? (BASELINE.json configs[4]: 100k-sample reads with a trace dump)"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flappie_amd import model as M
nfile, nsamp = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 100000
d = tempfile.mkdtemp(prefix="flappie_trace_")
M.write_mdl(os.path.join(d, "flipflop5_r941native.h"), M.synthetic_model(M.NET_LSTM5, 512, seed=1, ident="r941native"))
reads = os.path.join(d, "reads"); os.mkdir(reads)
rng = np.random.default_rng(1)
tool = os.path.join(ROOT, "flappie_amd", "fast5_tool")
for i in range(nfile):
    x = rng.normal(500, 60, nsamp); x[:300] = rng.normal(520, 4, 300)
    tmp = os.path.join(d, "r.i16"); np.clip(np.rint(x), 0, 8191).astype("<i2").tofile(tmp)
    subprocess.run([tool, "write", os.path.join(reads, "read_%05d.fast5" % i), "uuid-%05d" % i, "8192", "10", "1400", "4000", tmp], check=True)
env = dict(os.environ, FLAPPIE_MODEL_DIR=d, FLAPPIE_CLI_TIMING="1")
for tag, extra in (("no trace", []), ("--trace", ["--trace", os.path.join(d, "t.hdf5")]), ("--trace, no compression", ["--trace", os.path.join(d, "t0.hdf5"), "--hdf5-compression", "0"])):
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "flappie_amd", "flappie")] + os.environ.get("FLAPPIE_EXTRA", "").split() + extra + [reads], env=env, capture_output=True, text=True)
    dt = time.time() - t0
    print("%-26s rc %d  %.2f s  = %.1f Msamples/s (incl. %.1f s start-up)" % (tag, r.returncode, dt, nfile * nsamp / dt / 1e6, 0.9))
    print("   " + " | ".join(l.strip() for l in r.stderr.strip().split("\n")[-11:] if "s" in l)[:600])
    for f in ("t.hdf5", "t0.hdf5"):
        p = os.path.join(d, f)
        if os.path.exists(p): print("   %s: %.0f MB" % (f, os.path.getsize(p) / 1e6)); os.unlink(p)
