"""Search gains (input, recurrent, head, stay bias) of flappie_amd.model.synthetic_model per (kind, H) so that the
random model is INPUT-DRIVEN on N(0,1) reads yet not chaotic:

  * called bases per sample  >= 1/12
  * distinct 5-mers per 4000-sample read >= 100
  * a 1e-6 perturbation of the input moves the transition scores by < 5e-5  (model.py's criterion)

usage: python tools/dev/tune_gains.py KIND H [gi gs gf stay]     (no gains: grid search, prints the table)
TEST INFRASTRUCTURE: uses the oracle.
"""
import itertools
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flappie_amd import model as M  # noqa: E402


def measure(args):
    kind, H, gains, seed, nread, nsample = args
    from oracle import ffo
    mdl = M.synthetic_model(kind, H, seed=seed, gains=gains)
    om = ffo.OracleModel(mdl)
    rng = np.random.default_rng(1234)
    out = []
    with ffo.dot_mode(2):
        for r in range(nread):
            sig = rng.standard_normal(nsample).astype(np.float32)
            a = om.basecall(sig)
            pert = sig + np.float32(1e-6) * rng.standard_normal(nsample).astype(np.float32)
            tb = om.transitions(pert)
            sens = float(np.abs(tb - a["trans"]).max())
            s = a["basecall"]
            kmers = len({s[i:i + 5] for i in range(len(s) - 4)})
            comp = [s.count(c) / max(1, len(s)) for c in "ACGTZ"[: mdl.nbase]]
            out.append((len(s) / nsample, kmers, sens, min(comp), max(comp)))
    o = np.array(out)
    return gains, o[:, 0].mean(), o[:, 1].min(), o[:, 2].max(), o[:, 3].min(), o[:, 4].max()


def main():
    kind = {"lstm": M.NET_LSTM5, "grumod": M.NET_GRUMOD5, "rle": M.NET_LSTM5_RLE}[sys.argv[1]]
    H = int(sys.argv[2])
    nsample = int(os.environ.get("NSAMPLE", "4000"))
    seeds = [int(s) for s in os.environ.get("SEEDS", "1").split(",")]
    if len(sys.argv) > 3:
        grid = [tuple(float(x) for x in sys.argv[3:8])]
    else:
        gi = [float(x) for x in os.environ.get("GI", "3,5,8").split(",")]
        gs = [float(x) for x in os.environ.get("GS", "0.5,1.0,1.5,2.5").split(",")]
        gf = [float(x) for x in os.environ.get("GF", "4,10").split(",")]
        st = [float(x) for x in os.environ.get("STAY", "-0.5,0,0.6").split(",")]
        gc = [float(x) for x in os.environ.get("GC", "1").split(",")]
        grid = list(itertools.product(gi, gs, gf, st, gc))
    jobs = [(kind, H, g, s, 2, nsample) for g in grid for s in seeds]
    with ProcessPoolExecutor(int(os.environ.get("NPROC", "8"))) as ex:
        for (g, rate, kmers, sens, cmin, cmax), job in zip(ex.map(measure, jobs), jobs):
            ok = rate >= 1 / 12 and kmers >= 100 and sens < 5e-5
            print("%s H=%d seed=%d gains=%s  bases/sample %.4f (1 per %.1f)  5-mers %d  sens %.2e  comp %.2f..%.2f %s"
                  % (sys.argv[1], H, job[3], g, rate, 1 / max(rate, 1e-9), kmers, sens, cmin, cmax, "OK" if ok else ""), flush=True)


if __name__ == "__main__":
    main()
