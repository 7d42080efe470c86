#!/usr/bin/env python3
"""End-to-end throughput of the `flappie` binary on generated single-read fast5 files of mixed lengths
(fast5 read -> GPU signal preparation -> ragged batches -> FASTQ).  Development tool; run on the GPU box."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 384
counts = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["256", "1024"])]
readers = sys.argv[3].split(",") if len(sys.argv) > 3 else ["4"]      # --readers values to time (0 = the reader thread in the process)
d = tempfile.mkdtemp(prefix="flappie_cli_")
t0 = time.time()
grumod = len(sys.argv) > 4 and sys.argv[4] == "grumod"              # the r941_5mC model (GRUmod5, stride 2, 10 states) instead of r941_native
if grumod:
    M.write_mdl(os.path.join(d, "flipflop_r941native5mC.h"), M.synthetic_model(M.NET_GRUMOD5, hidden, seed=1, ident="r941native5mC"))
else:
    M.write_mdl(os.path.join(d, "flipflop5_r941native.h"), M.synthetic_model(M.NET_LSTM5, hidden, seed=1, ident="r941native"))
print("model file written in %.1f s" % (time.time() - t0), flush=True)
tool = os.path.join(ROOT, "flappie_amd", "fast5_tool")
nmax = max(counts)
reads = os.path.join(d, "reads")
os.mkdir(reads)
t0 = time.time()
# seeded noise around 500 +- 60 counts behind a 300-sample quiet stretch, 3500-5500 samples per file, written in one process (fast5_tool synth)
out = subprocess.run([tool, "synth", reads, str(nmax), "3500", "5500", "1"], check=True, capture_output=True, text=True).stdout.split()
total = [int(out[3]) / max(1, nmax)] * nmax
print("%d fast5 files written in %.1f s" % (nmax, time.time() - t0), flush=True)
env = dict(os.environ, FLAPPIE_MODEL_DIR=d)
for nr in readers:
    res = []
    for n in counts:
        env["FLAPPIE_CLI_TIMING"] = "1"
        wrap = os.environ.get("FLAPPIE_WRAP", "").split() if n == counts[-1] else []      # e.g. "rocprofv3 --kernel-trace --output-format csv -d DIR --" on the largest run
        dt = None
        for rep in range(int(os.environ.get("CLI_REPEATS", "3"))):                        # the best of a few runs: a single run's wall varies by +-5 %
            t0 = time.time()
            r = subprocess.run(wrap + [os.path.join(ROOT, "flappie_amd", "flappie")] + (["--model", "r941_5mC"] if grumod else []) + ["--readers", nr, "--limit", str(n), "-o", os.path.join(d, "out.fq"), reads], env=env,
                               capture_output=True, text=True)
            dt = min(dt, time.time() - t0) if dt is not None else time.time() - t0
        nrec = sum(1 for ln in open(os.path.join(d, "out.fq")) if ln.startswith("@uuid"))
        print("flappie --readers %s --limit %d: rc %d, %d records, %.2f s   %s" % (nr, n, r.returncode, nrec, dt, "\n" + r.stderr.strip()[-800:]), flush=True)
        res.append((n, dt))
    if len(res) >= 2:
        (n0, t0_), (n1, t1_) = res[0], res[-1]
        per_read = (t1_ - t0_) / (n1 - n0)
        print("readers %s: marginal cost %.3f ms per read (~%d samples) = %.2f Msamples/s; fixed cost %.1f s" % (nr, per_read * 1e3, int(np.mean(total)),
              np.mean(total) / per_read / 1e6, t0_ - n0 * per_read), flush=True)
