"""the flappie binary on the same 4096 generated fast5 files with different batch sizes / reader counts / models: the FASTQ must be
byte-identical whatever the batching (ragged batches, chunk-spanning pipeline, 512-read launches at H <= 256)"""
import hashlib, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flappie_amd import model as M
nfile = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tool = os.path.join(ROOT, "flappie_amd", "fast5_tool")
for hidden in (256, 384):
    d = tempfile.mkdtemp(prefix="flappie_soak_")
    M.write_mdl(os.path.join(d, "flipflop5_r941native.h"), M.synthetic_model(M.NET_LSTM5, hidden, seed=1, ident="r941native"))
    reads = os.path.join(d, "reads"); os.mkdir(reads)
    rng = np.random.default_rng(hidden)
    for i in range(nfile):
        n = int(rng.integers(1200, 6000))
        x = rng.normal(500, 60, n); x[:300] = rng.normal(520, 4, 300)
        tmp = os.path.join(d, "r.i16"); np.clip(np.rint(x), 0, 8191).astype("<i2").tofile(tmp)
        subprocess.run([tool, "write", os.path.join(reads, "read_%05d.fast5" % i), "uuid-%05d" % i, "8192", "10", "1400", "4000", tmp], check=True)
    env = dict(os.environ, FLAPPIE_MODEL_DIR=d)
    sums = {}
    # (round 6: the files through libhdf5 only, and every chunk in one-read-a-row batches, must give the same bytes as the defaults -- host/fast5_raw.c, packed batches)
    for tag, extra, dbg in (("default", [], ""), ("batch 64, 3 readers", ["--batch", "64", "--readers", "3"], ""), ("batch 200, 12 readers", ["--batch", "200", "--readers", "12"], ""),
                            ("batch 1024", ["--batch", "1024"], ""), ("libhdf5 reader", [], "hdf5_read"), ("one read a row", [], "no_pack"), ("in-process reader", ["--readers", "0"], "")):
        t0 = time.time()
        r = subprocess.run([os.path.join(ROOT, "flappie_amd", "flappie")] + extra + [reads], env=dict(env, FLAPPIE_DEBUG=dbg) if dbg else env, capture_output=True)
        assert r.returncode == 0, r.stderr[-500:]
        sums[tag] = hashlib.md5(r.stdout).hexdigest()
        print("H %d  %-24s %d records  md5 %s  %.1f s" % (hidden, tag, r.stdout.count(b"\n@uuid") + r.stdout.startswith(b"@uuid"), sums[tag], time.time() - t0), flush=True)
    assert len(set(sums.values())) == 1, sums
print("cli soak ok")
