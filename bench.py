#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s basecalled, r941_native shape, 256 reads x 4000 samples.

One "step" = one pass of the whole hot path (signal already resident in HBM -> convolutions ->
5 recurrent layers -> global-norm flip-flop scores -> forward/backward posterior -> Viterbi ->
base + quality strings + trace, plus the copy of the called strings back to the host) over one batch
of BASELINE.json's configs[1].  Reads are independent units: with N GPUs every rank runs the same
per-GPU workload on its own reads (weak scaling), with no data-path collective.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
SPLIT_PRODUCTS = 6                # bf16 MFMA products the split-bf16 layer kernel issues per fp32-exact product (ffhip_rnn_split.hip)
HIDDEN = 384                      # r941_native (flipflop5_202003) hidden size inferred from the model's size
NREAD, NSAMPLE = int(os.environ.get('FFHIP_BENCH_NREAD', '256')), 4000


def _cpu_worker(job):
    """One host core: the oracle's whole path over its own reads until the time budget is spent."""
    hidden, seed, nread, budget_s, max_reads = job
    from flappie_amd import model as M
    from oracle import ffo
    mdl = M.synthetic_model(M.NET_LSTM5, hidden, seed=1, ident="r941native")
    om = ffo.OracleModel(mdl)
    sig = np.random.default_rng(seed).standard_normal((max_reads, NSAMPLE)).astype(np.float32)
    t0 = time.time()
    n = 0
    while n < max_reads and (n == 0 or time.time() - t0 < budget_s):
        om.basecall(sig[n], want_trans=False)
        n += 1
    return n, time.time() - t0


def cpu_baseline(hidden, budget_s=12.0, max_reads=6):
    """The oracle (a scalar C port of the reference's algorithm, NOT the OpenBLAS reference itself, which cannot
    be built in this image) timed on the host cores of the GPU box: one single-threaded process per core, each
    over its own bounded sample of synthetic reads of the benchmark's shape, as the reference's README runs it
    (one flappie process per core under GNU parallel)."""
    import multiprocessing as mp
    try:
        ncore = len(os.sched_getaffinity(0))
    except AttributeError:
        ncore = os.cpu_count() or 1
    ncore = max(1, min(ncore, 64))
    jobs = [(hidden, 777 + k, NREAD, budget_s, max_reads) for k in range(ncore)]
    t0 = time.time()
    if ncore == 1:
        res = [_cpu_worker(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(ncore) as pool:       # spawn: the children must not inherit the HIP runtime
            res = pool.map(_cpu_worker, jobs)
    wall = time.time() - t0
    nread = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    return dict(value=round(nread * NSAMPLE / dt / 1e6, 6), unit="Msamples/s", cores=ncore, kind="port",
                sample="%d synthetic reads of %d samples over %d single-threaded processes (whole path), slowest worker %.1f s, %.1f s wall"
                       % (nread, NSAMPLE, ncore, dt, wall),
                per_core=round(nread * NSAMPLE / dt / 1e6 / ncore, 6))


def measured_traffic(hidden, rnn_path):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_traffic.json: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus
    WRITE_SIZE, separate passes).  Counters cannot be read inside this process, so the value is the
    profile's, keyed by shape and kernel; null when no profile of this shape and kernel is committed."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        for e in (t if isinstance(t, list) else [t]):
            if e.get("hidden") == hidden and e.get("nread") == NREAD and e.get("nsample") == NSAMPLE and e.get("rnn_path", 2 if e.get("fused") else 1) == rnn_path:
                return e.get("recurrent_layer_hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("FFHIP_INFLIGHT", "1")),
                    help="batches in flight per GPU (each on its own HIP stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hidden", type=int, default=HIDDEN)
    args = ap.parse_args()

    # Only the JSON line may reach stdout: RCCL prints a version banner there at init, so fd 1 is pointed at
    # stderr for the duration of the run and the result is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    # FFHIP_BENCH_FORCE_DIST=1: take the RCCL code path (init, barrier, MAX all-reduce) with a single rank too --
    # lets a 1-GPU box exercise exactly what the N > 1 launches run
    dist_on = world > 1 or bool(os.environ.get("FFHIP_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)

    from flappie_amd import binding as B
    from flappie_amd import model as M

    eng = B.Engine(local_rank)
    mdl = M.synthetic_model(M.NET_LSTM5, args.hidden, seed=1, ident="r941native")
    dm = B.DeviceModel(eng, mdl)
    rng = np.random.default_rng(20260928 + rank)
    sig = rng.standard_normal((NREAD, NSAMPLE)).astype(np.float32)
    nfl = max(1, min(args.inflight, 2))
    batches = [B.Batch(dm, NREAD, NSAMPLE) for _ in range(nfl)]
    for b in batches:
        b.set_signals(sig)               # inputs resident in HBM before the timed region

    def barrier():
        if dist_on:
            dist.barrier()
        eng.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def run_steps(n):
        pending = []
        for i in range(n):
            b = batches[i % nfl]
            if len(pending) == nfl:
                pending.pop(0).finish()
            b.run(1.0, 0)
            pending.append(b)
        for b in pending:
            b.finish()

    run_steps(args.warmup)
    eng.set_profiling(True)              # HIP events on the kernels' own stream; no host synchronisation added
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    prof = [b.profile() for b in batches]
    eng.set_profiling(False)

    if dist_on:       # MAX over ranks of the timed region (flappie_amd/shard.py::max_over_ranks, inlined so that it also runs at world 1)
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        nblock = batches[0].nblock
        value = world * args.steps * NREAD * NSAMPLE / dt / 1e6
        # dominant kernel: one recurrent layer.  Algorithmic work per read per block:
        # 2*H*4H FLOP for the recurrence, and the same again for the input projection when the layer
        # runs fused (projection and recurrence in one persistent launch; `inproj` then has 0 launches).
        rec = prof[-1]["recurrent"]
        fused = prof[-1]["inproj"]["launches"] == 0
        rnn_path = batches[-1].rnn_path()
        flop_layer = (2.0 if fused else 1.0) * 2.0 * args.hidden * 4 * args.hidden * NREAD * nblock
        launches_per_layer = rec["launches"] / 5.0
        ms_layer = rec["ms"] / 5.0
        achieved = flop_layer / (ms_layer * 1e-3) / 1e12
        if rnn_path == 3:
            # fp32-exact products out of bf16 MFMAs: each algorithmic (fp32) multiply-add is six bf16 MFMA products
            # (three-way split of both operands, the three smallest cross terms dropped).  `achieved` counts the
            # ALGORITHMIC fp32 FLOPs; the ceiling of this formulation is the dense bf16 peak / 6.
            kname = "k_lstm_split<%d> (input projection + recurrence of one layer on bf16 MFMAs over 3-way split operands, %d dependent steps)" % (args.hidden // 128, nblock)
            peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS
            peak_note = ("dense bf16 MFMA peak %.0f TFLOP/s / %d products per fp32-exact product; the kernel issues %.1f TFLOP/s of bf16 MFMA work "
                         "= %.3f of the bf16 peak; against the f32-input MFMA peak (%.1f) the algorithmic rate is %.3f"
                         % (PEAK_BF16_MFMA_TFLOPS, SPLIT_PRODUCTS, achieved * SPLIT_PRODUCTS, achieved * SPLIT_PRODUCTS / PEAK_BF16_MFMA_TFLOPS,
                            PEAK_F32_MFMA_TFLOPS, achieved / PEAK_F32_MFMA_TFLOPS))
            dtype = "f32 (products as 6 bf16 MFMA terms over 3-way split operands, f32 accumulate; gate math f32)"
        else:
            kname = ("k_lstm_fused (input projection + recurrence of one layer, %d dependent steps)" if fused
                     else "k_rnn_persist (recurrence of one layer, %d dependent steps)") % nblock
            peak = PEAK_F32_MFMA_TFLOPS
            peak_note = "f32-input MFMA peak (v_mfma_f32_16x16x4_f32)"
            dtype = "f32"
        out = {
            "metric": "Msamples/s basecalled (r941_native, 4k-sample chunks)",
            "value": round(value, 4), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic (seeded N(0,1) signal, seeded random-init weights of the r941_native architecture)",
            "config": {"workload": "r941_native-shape LSTM5 H=%d, batch=256 synthetic 4000-sample reads per GPU, "
                                   "posterior decode + trace (BASELINE.json configs[1])" % args.hidden,
                       "reads_per_step": NREAD, "samples_per_read": NSAMPLE, "blocks_per_read": nblock,
                       "batches_in_flight": nfl, "parallelism": "reads sharded by rank, no collective"},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": round(achieved, 3), "peak": round(peak, 2), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": measured_traffic(args.hidden, rnn_path),
                         "peak_note": peak_note,
                         "flop_per_launch": flop_layer / launches_per_layer,
                         "avg_launch_ms": round(ms_layer / launches_per_layer, 6),
                         "launches_per_layer": launches_per_layer},
            "kernel_ms_per_step": {k: round(v["ms"], 4) for k, v in prof[-1].items()},
            # decode side (posterior + Viterbi + assembly + trace): algorithmic bytes per block (SURVEY.md section 8d:
            # 4P read + nstate traceback + 8 path/qpath, plus 4P read + 4P write for the posterior) against HBM peak.
            # At 256 reads these kernels are latency-bound chains, not bandwidth-bound.
            "decode_hbm": (lambda ms, nbytes: {"achieved": round(nbytes / (ms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                                               "bytes_per_block": 12 * mdl.nparam + mdl.nstate + 8})(
                prof[-1]["posterior"]["ms"] + prof[-1]["viterbi_assembly"]["ms"],
                float(NREAD) * nblock * (12 * mdl.nparam + mdl.nstate + 8)),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.hidden)
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    for b in batches:
        b.close()
    dm.close()
    eng.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
