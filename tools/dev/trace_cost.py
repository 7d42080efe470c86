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
tool = os.path.join(ROOT, "flappie_amd", "fast5_tool")
t0 = time.time()
subprocess.run([tool, "synth", reads, str(nfile), str(nsamp), str(nsamp + 1), "1"], check=True, capture_output=True)      # seeded noise behind a quiet stretch, one process
print("%d fast5 files of %d samples written in %.1f s" % (nfile, nsamp, time.time() - t0), flush=True)
env = dict(os.environ, FLAPPIE_MODEL_DIR=d, FLAPPIE_CLI_TIMING="1")
res = {}
for tag, extra in (("no trace", []), ("--trace", ["--trace", os.path.join(d, "t.hdf5")]), ("--trace, no compression", ["--trace", os.path.join(d, "t0.hdf5"), "--hdf5-compression", "0"])):
    for n in (nfile // 2, nfile):
        t0 = time.time()
        r = subprocess.run([os.path.join(ROOT, "flappie_amd", "flappie")] + os.environ.get("FLAPPIE_EXTRA", "").split() + extra + ["--limit", str(n), "-o", os.path.join(d, "out.fq"), reads],
                           env=env, capture_output=True, text=True)
        dt = time.time() - t0
        res[(tag, n)] = dt
        print("%-26s %5d files: rc %d  %.2f s  = %.1f Msamples/s start-up included" % (tag, n, r.returncode, dt, n * nsamp / dt / 1e6), flush=True)
        print("   " + " | ".join(l.strip() for l in r.stderr.strip().split("\n")[-12:] if "s" in l)[:700])
        for f in ("t.hdf5", "t0.hdf5"):
            p = os.path.join(d, f)
            if os.path.exists(p): print("   %s: %.0f MB" % (f, os.path.getsize(p) / 1e6)); os.unlink(p)
    print("%-26s marginal rate between %d and %d files: %.1f Msamples/s" % (tag, nfile // 2, nfile, (nfile - nfile // 2) * nsamp / (res[(tag, nfile)] - res[(tag, nfile // 2)]) / 1e6), flush=True)
