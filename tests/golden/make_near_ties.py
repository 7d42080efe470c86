#!/usr/bin/env python3
"""tests/golden/near_ties.npz: reads on which a parity campaign found the GPU path and the oracle calling DIFFERENT base strings on
transition scores that agree to rounding -- near-ties of the posterior decode.  They are recorded so that a
campaign can tell a known read from a new one, and tests/test_fuzz_tail_gpu.py::test_recorded_near_tie_reads
re-runs them.

  0: profiles/r03_parity_packed.txt -- tools/parity_pack.py 1024 3000 (round 3): GRUmod H = 256, synthetic_model(seed=7) under the
     round-3 gains, read 99 of the batch (2743 samples): 222 bases against the oracle's 220, |dtrans| 7.6e-6, paths apart in blocks
     246-248.  The signal is regenerated exactly as that tool drew it.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import model as M  # noqa: E402

out = {}
# read 0: parity_pack.py's second model (GRUmod, H 256, seed 7), nread 1024, tmax 3000
nread, tmax, seed = 1024, 3000, 7
rng = np.random.default_rng(200 + seed)
lens = np.sort(rng.integers(300, tmax + 1, nread))[::-1]
sigs = [rng.standard_normal(int(n)).astype(np.float32) for n in lens]
assert sigs[99].size == 2743
out.update(kind0=np.array(M.NET_GRUMOD5), hidden0=np.array(256), model_seed0=np.array(seed), gains0=np.array(M.SYNTH_GAINS_R3[M.NET_GRUMOD5], dtype=np.float64),
           signal0=sigs[99], source0=np.array("profiles/r03_parity_packed.txt read 99"), blocks_apart0=np.array(3))
out["n"] = np.array(1)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "near_ties.npz"), **out)
print("near_ties.npz: %d reads" % int(out["n"]))
