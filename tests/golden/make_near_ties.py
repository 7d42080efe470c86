#!/usr/bin/env python3
"""tests/golden/near_ties.npz: reads on which a parity campaign found the GPU path and the oracle calling DIFFERENT base strings on
transition scores that agree to rounding -- near-ties of the posterior decode.  They are recorded so that a
campaign can tell a known read from a new one, and tests/test_fuzz_tail_gpu.py::test_recorded_near_tie_reads
re-runs them.

  0: profiles/r03_parity_packed.txt -- tools/parity_pack.py 1024 3000 (round 3): GRUmod H = 256, synthetic_model(seed=7) under the
     round-3 gains, read 99 of the batch (2743 samples): 222 bases against the oracle's 220, |dtrans| 7.6e-6, paths apart in blocks
     246-248.  The signal is regenerated exactly as that tool drew it.
  1: profiles/r04_parity_h384.txt -- tools/parity_h384.py 2048 2500 (round 4, input-driven models): LSTM H = 384, bench.py's model
     (seed 1, round-4 gains), batch 5 read 68 (2500 samples): 457 bases against 457, one base apart, |dtrans| 1.3e-5 -- the one read of
     2048 that the engine calls differently from BOTH the scalar oracle and the oracle through OpenBLAS (which differ from each other on
     two more reads of that campaign).
  2: profiles/r04_parity_h384.txt -- tools/parity_h384.py 8192 2500 c2 on the round's FINAL tree (weights-stationary convolution, split head): same
     model, batch 4 read 148 (2500 samples): 462 bases against 462, three apart, |dtrans| 1.3e-5 -- the one read of 8192 / 3.36 M bases called
     differently from both evaluations of the reference algorithm (read 1 above no longer is: the scores moved in the last bits with the kernels).
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
           signal0=sigs[99], source0=np.array("profiles/r03_parity_packed.txt read 99"), blocks_apart0=np.array(3), dtrans_bound0=np.array(1e-5), f32_equals_oracle0=np.array(1))
# read 1: parity_h384.py's batches (rng 3840; pairs of uniform batches and pairs of ragged ones alternate), batch 5 read 68
rng = np.random.default_rng(3840)
tmax = 2500
batches = []
for k in range(6):
    if (k // 2) % 2 == 0:
        batches.append([rng.standard_normal(tmax).astype(np.float32) for _ in range(256)])
    else:
        lens = np.sort(rng.integers(1500, tmax + 1, 256))[::-1]
        batches.append([rng.standard_normal(int(n)).astype(np.float32) for n in lens])
out.update(kind1=np.array(M.NET_LSTM5), hidden1=np.array(384), model_seed1=np.array(1), gains1=np.array(M.SYNTH_GAINS[M.NET_LSTM5], dtype=np.float64),
           signal1=batches[5][68], source1=np.array("profiles/r04_parity_h384.txt batch 5 read 68"), blocks_apart1=np.array(4), dtrans_bound1=np.array(2e-5), f32_equals_oracle1=np.array(-1))
out.update(kind2=np.array(M.NET_LSTM5), hidden2=np.array(384), model_seed2=np.array(1), gains2=np.array(M.SYNTH_GAINS[M.NET_LSTM5], dtype=np.float64),
           signal2=batches[4][148], source2=np.array("profiles/r04_parity_h384.txt (8192 reads) batch 4 read 148"), blocks_apart2=np.array(48), dtrans_bound2=np.array(2e-5), f32_equals_oracle2=np.array(-1))
out["n"] = np.array(3)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "near_ties.npz"), **out)
print("near_ties.npz: %d reads" % int(out["n"]))
