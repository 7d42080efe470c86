#!/usr/bin/env python3
"""Statistical parity campaign on the GPU box: many random reads of random lengths through ragged batches against the
oracle, counting base-string / quality / path mismatches and the largest transition-score difference.  The engine's
dot products are summed in a different order than the oracle's (and the reference's OpenBLAS), so identical calls are
an empirical property; this measures how often it fails."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import binding as B  # noqa: E402
from flappie_amd import model as M  # noqa: E402
from oracle import ffo  # noqa: E402

nread = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = B.Engine(0)
tot = dict(reads=0, bases=0, base_mismatch=0, qual_mismatch=0, path_mismatch=0, qual_chars_diff=0, worst=0.0)
t0 = time.time()
models = ((M.NET_LSTM5, 96, 1), (M.NET_LSTM5, 64, 2), (M.NET_GRUMOD5, 64, 3), (M.NET_LSTM5, 128, 4))
if len(sys.argv) > 2 and sys.argv[2] == "split":        # only shapes the split-bf16 layer kernel takes (H = 128, 256)
    models = ((M.NET_LSTM5, 128, 4), (M.NET_LSTM5, 128, 5), (M.NET_LSTM5, 256, 6))
# "inflight" as the third argument: every measured batch runs BESIDE another batch's layer kernels (H = 256, long reads) --
# the mode bench.py and the flappie binary use; DESIGN.md section 5.4 is why this is worth a campaign of its own
companion = None
if len(sys.argv) > 3 and sys.argv[3] == "inflight":
    cmdl = M.synthetic_model(M.NET_LSTM5, 256, seed=77)
    cdm = B.DeviceModel(eng, cmdl)
    companion = B.Batch(cdm, 256, 12000)
    companion.set_signals(np.random.default_rng(5).standard_normal((256, 12000)).astype(np.float32))
for kind, H, seed in models:
    mdl = M.synthetic_model(kind, H, seed=seed)
    om = ffo.OracleModel(mdl)
    dm = B.DeviceModel(eng, mdl)
    rng = np.random.default_rng(100 + seed)
    lens = np.sort(rng.integers(300, 2500, nread))[::-1]
    sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
    b = B.Batch(dm, nread, int(lens.max()))
    b.set_signals_ragged(sigs)
    if companion is not None: companion.run(1.0, B.RUN_NO_DECODE)
    b.run(); b.finish()
    if companion is not None: companion.finish()
    for r, x in enumerate(sigs):
        ref = om.basecall(x)
        tot["reads"] += 1
        tot["bases"] += len(ref["basecall"])
        tot["worst"] = max(tot["worst"], float(np.abs(b.transitions(r) - ref["trans"]).max()))
        if b.basecall(r) != ref["basecall"]:
            tot["base_mismatch"] += 1
        elif b.quality(r) != ref["quality"]:
            tot["qual_mismatch"] += 1
            tot["qual_chars_diff"] += sum(1 for a, c in zip(b.quality(r), ref["quality"]) if a != c)
        if not np.array_equal(b.path(r)[0], ref["path"]):
            tot["path_mismatch"] += 1
    b.close(); dm.close()
    print("kind %d H %d done (%.0f s): %s" % (kind, H, time.time() - t0, tot), flush=True)
print("campaign:", tot)
