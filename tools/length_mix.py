#!/usr/bin/env python3
"""What a nanopore-like read-length mix costs through the `flappie` binary's batcher (config 3's workload).

The reference takes reads of any length one at a time (flappie.c:245-262, 334-385); here a batch costs what its longest read costs, so the
batcher's grouping decides how much of the GPU a mixed directory gets.  Two directories of generated single-read fast5 files with about the
same number of samples:
   uniform   3500-5500 raw samples a file (bench.py's host-fed leg)
   mixed     log-normal lengths, median 8000, sigma 1, clipped to 1000 .. 200 000 samples (fast5_tool synthln)
each through `flappie --readers R` at --limit N/2 and N; the rate is MARGINAL (long run minus short run, as bench.py's host_fed: the batch objects of a
packed run -- up to 90 GB each at 384 hidden units -- are allocated once, in the first chunks), the
padding efficiency is the binary's own account (FLAPPIE_CLI_TIMING: samples / (slots x longest read) by batch and by 16-read tile).
Run on the GPU box.   usage: tools/length_mix.py [hidden=384] [nfiles=65536] [readers=4] [env NAME=VALUE ...: extra environment for the mixed runs]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 384
nfiles = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
readers = sys.argv[3] if len(sys.argv) > 3 else "4"
extra = dict(a.split("=", 1) for a in sys.argv[4:])
base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
d = tempfile.mkdtemp(prefix="ffhip_lenmix_", dir=base)
exe, tool = os.path.join(ROOT, "flappie_amd", "flappie"), os.path.join(ROOT, "flappie_amd", "fast5_tool")
try:
    M.write_mdl(os.path.join(d, "flipflop5_r941native.h"), M.synthetic_model(M.NET_LSTM5, hidden, seed=1, ident="r941native"))
    mixes = {}
    os.mkdir(os.path.join(d, "mixed"))
    out = subprocess.run([tool, "synthln", os.path.join(d, "mixed"), str(nfiles), "8000", "1.0", "1000", "200000", "20260929"], check=True, capture_output=True, text=True).stdout.split()
    mixes["mixed"] = (nfiles, int(out[3]))
    n_uni = min(32768, max(1024, int(out[3]) // 4500 // 256 * 256))
    os.mkdir(os.path.join(d, "uniform"))
    out = subprocess.run([tool, "synth", os.path.join(d, "uniform"), str(n_uni), "3500", "5500", "20260929"], check=True, capture_output=True, text=True).stdout.split()
    mixes["uniform"] = (n_uni, int(out[3]))
    print("H = %d, --readers %s; mixed: %d files, %.1f Msamples (log-normal, median 8000, sigma 1, 1000 .. 200 000); uniform: %d files, %.1f Msamples (3500 .. 5500)"
          % (hidden, readers, nfiles, mixes["mixed"][1] / 1e6, n_uni, mixes["uniform"][1] / 1e6), flush=True)
    for name in ("uniform", "mixed"):
        n = mixes[name][0]
        env = dict(os.environ, FLAPPIE_MODEL_DIR=d, FLAPPIE_CLI_TIMING="1")
        if name == "mixed":
            env.update(extra)
        res = {}
        # (a first short run is thrown away: what the allocation of the 2 x 100 GB packed batch objects costs depends on what the box's memory has been through -- 0 .. 5 s,
        # the first process to touch it pays most --; the timed runs then all start from the same state)
        subprocess.run([exe, "--readers", readers, "--limit", str(n // 2), "-o", os.path.join(d, "out.fq"), os.path.join(d, name)], env=env, capture_output=True, text=True)
        for rep in range(2):
            for lim in (n // 2, n):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "--readers", readers, "--limit", str(lim), "-o", os.path.join(d, "out.fq"), os.path.join(d, name)], env=env, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                called = [ln for ln in r.stderr.splitlines() if ln.startswith("basecalled:")]
                pad = [ln for ln in r.stderr.splitlines() if ln.startswith("batches:")]
                raw = int(called[-1].split()[7]) if called else 0
                nrd = int(called[-1].replace(",", " ").split()[1]) if called else 0
                if r.returncode != 0 or nrd != lim:
                    print("  %s --limit %d: rc %d, %d reads\n%s" % (name, lim, r.returncode, nrd, r.stderr[-1500:]), flush=True)
                if lim not in res or dt < res[lim][0]:
                    res[lim] = (dt, raw, pad[-1] if pad else "")
        (t0_, r0, _), (t1_, r1, pad) = res[n // 2], res[n]
        print("%-8s %6d files: %.2f s, %6d files: %.2f s -> marginal %.1f Msamples/s (whole long run: %.1f)   %s%s"
              % (name, n // 2, t0_, n, t1_, (r1 - r0) / (t1_ - t0_) / 1e6, r1 / t1_ / 1e6, pad, ("   [" + " ".join("%s=%s" % kv for kv in extra.items()) + "]") if extra and name == "mixed" else ""), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
