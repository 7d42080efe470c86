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
SPLIT_PRODUCTS = 3                # fp16 MFMA products the split layer kernels issue per fp32 multiply-add block (two fp16 slices per operand,
                                  # w0x0 + w0x1 + w1x0: flappie_amd/csrc/ffhip_split.hpp; the -DFFHIP_SPLIT_BF16X3 build issues 6)
HIDDEN = 384                      # r941_native (flipflop5_202003) hidden size inferred from the model's size

# Workloads.  c2 is the headline (BASELINE.json configs[1], the configuration the metric is quoted on) and the default;
# the others let the driver or a reader reproduce the figures DESIGN.md quotes for configs[3] / configs[4] and the
# smaller r941_native file with the same JSON line (own roofline, own CPU leg).  kind: 0 LSTM5, 1 GRUmod5, 2 LSTM5 + run-length head.
CONFIGS = {
    "c2":   dict(kind=0, hidden=384, nread=256, nsample=4000, steps=200, warmup=5, inflight=2, pair=1, ident="r941native",
                 metric="Msamples/s basecalled (r941_native, 4k-sample chunks)",
                 label="r941_native-shape LSTM5 H=384, batch=256 synthetic 4000-sample reads per GPU, posterior decode + trace (BASELINE.json configs[1])"),
    "h256": dict(kind=0, hidden=256, nread=1024, nsample=4000, steps=100, warmup=5, inflight=2, ident="r941native",
                 metric="Msamples/s basecalled (r941_native 20200220-size model, 4k-sample chunks)",
                 label="r941_native-shape LSTM5 H=256 (the 41.8 MB model file), batch=1024 synthetic 4000-sample reads per GPU (what one layer launch of the packed form takes at H = 256: 16 members a group, two workgroups per CU), posterior decode + trace"),
    "c4":   dict(kind=1, hidden=256, nread=1024, nsample=4000, steps=40, warmup=3, inflight=2, ident="r941_5mC",
                 metric="Msamples/s basecalled (r941_5mC, 4k-sample chunks)",
                 label="r941_5mC-shape GRUmod5 H=256, stride 2 (2000 blocks per read), 10 flip-flop states, batch=1024 synthetic 4000-sample reads per GPU (what one layer launch of the packed GRUmod form takes: "
                       "16 members a group, two workgroups per CU), "
                       "posterior decode + trace (BASELINE.json configs[3])"),
    "c5":   dict(kind=0, hidden=512, nread=256, nsample=100000, steps=6, warmup=2, inflight=2, ident="r103native",
                 metric="Msamples/s basecalled (r103_native standing in for r10C_pcr, 100k-sample reads, trace on)",
                 label="r103_native-shape LSTM5 H=512 (SURVEY.md section 0.3: there is no r10C_pcr model), batch=256 synthetic 100000-sample reads per GPU, "
                       "posterior decode + trace (BASELINE.json configs[4])"),
    "rle":  dict(kind=2, hidden=384, nread=256, nsample=4000, steps=100, warmup=3, inflight=2, pair=1, ident="rle_r941native",
                 metric="Msamples/s run-length called (rle_r941_native, 4k-sample chunks)",
                 label="rle_r941_native-shape LSTM5 H=384 + run-length head (runnie), batch=256 synthetic 4000-sample reads per GPU"),
}
SURVEY_OPENBLAS_PER_CORE = {(0, 384): 0.010, (0, 256): 0.035, (0, 512): 0.0046, (1, 256): 0.0185}   # Msamples/s/core, SURVEY.md section 6 [probe]


def _oracle_with_blas(mode):
    """the oracle library in dot mode `mode`; mode 3 (OpenBLAS) falls back to 2 (own vectorised kernels) when no library loads"""
    from oracle import ffo
    return ffo.use_dot_mode(mode)


def effective_cpus():
    """(CPUs this process may actually use, hardware threads it can be scheduled on, why): the affinity mask says where, the cgroup's CPU
    bandwidth quota how much -- on the GPU boxes of this pool a container sees 256 hardware threads and is granted 16 CPUs' worth of time
    (cpu.max `1600000 100000`); processes beyond the quota only throttle each other."""
    try:
        naff = len(os.sched_getaffinity(0))
    except AttributeError:
        naff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                t = fh.read().split()
            if path.endswith("cpu.max"):
                if t and t[0] != "max":
                    quota = float(t[0]) / float(t[1])
            else:
                q = float(t[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    p = float(fh.read().split()[0])
                if q > 0:
                    quota = q / p
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is None or quota >= naff:
        return naff, naff, "affinity mask"
    return max(1, int(quota + 0.5)), naff, "cgroup CPU quota %.1f of %d hardware threads" % (quota, naff)


def _cpu_worker(job):
    """One host core: the oracle's whole path over its own reads until the time budget is spent."""
    kind, hidden, ident, nsample, seed, budget_s, max_reads, mode = job
    os.environ["OPENBLAS_NUM_THREADS"] = "1"          # README.md:76-79: one thread per process
    from flappie_amd import model as M
    from oracle import ffo
    mdl = M.synthetic_model(kind, hidden, seed=1, ident=ident)
    om = ffo.OracleModel(mdl)
    _oracle_with_blas(mode)
    sig = np.random.default_rng(seed).standard_normal((max_reads, nsample)).astype(np.float32)
    call = om.runlength_call if kind == M.NET_LSTM5_RLE else (lambda x: om.basecall(x, want_trans=False))
    t0 = time.time()
    n = 0
    while n < max_reads and (n == 0 or time.time() - t0 < budget_s):
        call(sig[n])
        n += 1
    return n, time.time() - t0


def cpu_baseline(cfg, budget_s=12.0):
    """The oracle's algorithm (a C port of the reference's, NOT the OpenBLAS reference itself, which cannot be built in this
    image) timed on the host cores of the GPU box, the way the reference's README runs flappie: one single-threaded process
    per core under GNU parallel, each over its own bounded sample of synthetic reads of the benchmark's shape.  Dot products
    run through oracle/cpu_ref.c's vectorised kernels (AVX-512 / AVX2 clones: sgemv for the recurrent steps, a register-tiled
    sgemm for projections and convolutions) -- the shapes OpenBLAS would run, within ~1.5-2x of the survey's OpenBLAS probe
    on one core; the reference-order scalar oracle that the parity tests use is ~5x slower and is not what is timed."""
    import multiprocessing as mp
    ncore, nhw, why = effective_cpus()          # every CPU this container is GRANTED, as the reference's README runs it (one process per core under GNU parallel)
    nsample = min(cfg["nsample"], 20000)        # long-read configs: a 20 000-sample prefix per read keeps the sample bounded
    max_reads = max(2, int(400000 // nsample))
    mode, blas = _oracle_with_blas(3)                 # 3 = the GEMV / GEMM calls go to a real OpenBLAS when this host has one
    jobs = [(cfg["kind"], cfg["hidden"], cfg["ident"], nsample, 777 + k, budget_s, max_reads, mode) for k in range(ncore)]
    one = _cpu_worker((cfg["kind"], cfg["hidden"], cfg["ident"], nsample, 776, 3.0, 2, mode))      # one core alone, nothing competing for memory bandwidth
    own = _cpu_worker((cfg["kind"], cfg["hidden"], cfg["ident"], nsample, 776, 3.0, 2, 2)) if mode == 3 else one      # the same on the port's own kernels
    t0 = time.time()
    if ncore == 1:
        res = [_cpu_worker(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(ncore) as pool:       # spawn: the children must not inherit the HIP runtime
            res = pool.map(_cpu_worker, jobs)
    wall = time.time() - t0
    nread = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    return dict(value=round(nread * nsample / dt / 1e6, 6), unit="Msamples/s", cores=ncore, kind="port+openblas" if blas else "port",
                sample="%d synthetic reads of %d samples over %d single-threaded processes (whole path: oracle algorithm, %s), slowest worker %.1f s, %.1f s wall"
                       % (nread, nsample, ncore, ("GEMV / GEMM through %s [%s], one thread each; element-wise loops oracle/cpu_ref.c" % blas) if blas else
                          "vectorised kernels of oracle/cpu_ref.c", dt, wall),
                per_core=round(nread * nsample / dt / 1e6 / ncore, 6),
                hardware_threads=nhw, cores_limited_by=why,
                whole_host_if_linear=round(nread * nsample / dt / 1e6 / ncore * nhw, 4),
                one_core_alone=round(one[0] * nsample / one[1] / 1e6, 6),
                one_core_alone_own_kernels=round(own[0] * nsample / own[1] / 1e6, 6),
                blas_library=blas[0] if blas else None,
                reference_openblas_per_core_survey=SURVEY_OPENBLAS_PER_CORE.get((cfg["kind"] % 2, cfg["hidden"])),
                note=("kind=port+openblas: the reference's layers.c / flappie_matrix.c cannot be built in this image (no <cblas.h>); this is the oracle's restatement of "
                      "them calling the SAME cblas_sgemv / cblas_sgemm shapes (layers.c:1009, :250, flappie_matrix.c:384) in an OpenBLAS found on this host, dlopen()ed. "
                      if blas else "kind=port: no LP64 OpenBLAS found on this host; the port's own vectorised kernels. ") +
                     "cores = the CPUs this container is granted (cgroup quota; rounds 1-3 started 64 processes under a 16-CPU quota and read the throttling as memory-bandwidth contention); "
                     "per_core = value / cores; one_core_alone = one process alone; whole_host_if_linear = per_core x the host's hardware threads (an extrapolation, not a measurement); "
                     "reference_openblas_per_core_survey = the survey container's probe of the real reference (SURVEY.md section 6, other host)")


def measured_traffic(name, cfg, rnn_path, paired=False):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*traffic*.json:
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE, separate passes).  Counters cannot
    be read inside this process, so the value is the profile's, keyed by shape and kernel; null when no profile of this
    shape and kernel is committed.  The newest round's file wins."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*traffic.json"))):
        try:
            with open(path) as fh:
                t = json.load(fh)
        except (OSError, ValueError):
            continue
        for e in (t if isinstance(t, list) else [t]):
            if (e.get("hidden") == cfg["hidden"] and e.get("nread") == cfg["nread"] and e.get("nsample") == cfg["nsample"]
                    and e.get("kind", 0) == cfg["kind"] and e.get("rnn_path", 2 if e.get("fused") else 1) == rnn_path
                    and e.get("reads_per_launch", cfg["nread"]) == cfg["nread"] * (2 if paired else 1)):
                best = (e.get("recurrent_layer_hbm_bytes_per_launch"), "profiles/" + os.path.basename(path))
    return best if best else (None, None)


def self_launch(ngpus, argv):
    """`python bench.py --gpus N` started by hand (no WORLD_SIZE in the environment): become the launcher the driver would have
    been -- one rank per GPU under torch.distributed.run on 127.0.0.1 -- instead of asserting.  Returns only on exec failure."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


class _StubBatch:
    """FFHIP_BENCH_STUB=1 (CPU tests of the launch / timing / reduction plumbing): stands where a batch of the HIP engine stands."""
    nblock = 800

    def __init__(self, *a):
        pass

    def set_signals(self, sig):
        pass

    def run(self, temperature, flags):
        time.sleep(0.002)

    def finish(self):
        pass

    def profile(self):
        z = {"ms": 1.0, "launches": 5}
        return {"conv": dict(z), "inproj": {"ms": 0.0, "launches": 0}, "recurrent": dict(z), "head_crf": dict(z), "posterior": dict(z), "viterbi_assembly": dict(z)}

    def rnn_path(self):
        return 3

    def close(self):
        pass


class _StubEngine:
    def __init__(self, *a):
        pass

    def synchronize(self):
        pass

    def set_profiling(self, on):
        pass

    def close(self):
        pass


def host_fed_leg(cfg, rank, local_rank, world, dist, nfiles=None):
    """The path that feeds a GPU in production, per rank, beside `value` (never as it): the `flappie` BINARY over this rank's shard of a
    directory of single-read fast5 files -- reader child processes -> GPU signal preparation -> ragged batches -> FASTQ (README.md:81-83
    runs one flappie per core with GNU parallel; here one per GPU, `--shard rank/world`, no inter-process traffic; flappie.c:334-385).
    Readers per process = what the host's cores allow per GPU.  The rate is MARGINAL (a long run minus a short run over the same
    shard), so the fixed start-up (model text parse, HIP start, first allocations) is not in it; whole-job = all ranks' samples / the
    slowest rank's time.  Returns a dict for rank 0, None elsewhere; {"skipped": why} when the binary or libhdf5 is missing."""
    import shutil
    import subprocess
    import tempfile
    from flappie_amd import model as M
    exe, tool = os.path.join(ROOT, "flappie_amd", "flappie"), os.path.join(ROOT, "flappie_amd", "fast5_tool")
    if cfg["kind"] != M.NET_LSTM5 or not (os.path.exists(exe) and os.path.exists(tool)):
        return {"skipped": "needs the flappie binary + fast5_tool (libhdf5 at build time) and an LSTM5 flip-flop model"} if rank == 0 else None
    ncore = effective_cpus()[0]                 # what the container is granted, not what it can see
    # reader processes per rank: one reads ~50 Msamples/s of single-read files, and every reader beyond what the GPU consumes costs CPU for nothing (twelve of them:
    # 0.026 CPU-s per million samples and 108 Msamples/s; two to four: 0.016 and 111-114 -- profiles/r05_host_scaling.txt).  Eight ranks inside a 16-CPU grant: two each.
    readers = max(1, min(4, ncore // max(1, world)))
    nfiles = nfiles or int(os.environ.get("FFHIP_BENCH_HOSTFED_FILES", "32768"))      # per rank (the short run is a quarter of it: a steady-state marginal rate needs ~1 s of work)
    n_short = max(512, nfiles // 4)
    obj = [None]
    if rank == 0:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        obj[0] = tempfile.mkdtemp(prefix="ffhip_hostfed_", dir=base)
        os.mkdir(os.path.join(obj[0], "reads"))
    if dist is not None:
        dist.broadcast_object_list(obj, src=0)
    d = obj[0]
    out = None
    try:
        t0 = time.time()
        # every rank fills its own stripe of the shared directory (file index = rank + k * world: exactly the files `--shard rank/world` takes)
        gen = subprocess.run([tool, "synth", os.path.join(d, "reads"), str(nfiles), "3500", "5500", "20260928", str(rank), str(world)], capture_output=True, text=True)
        if rank == 0:
            M.write_mdl(os.path.join(d, "flipflop5_r941native.h"), M.synthetic_model(cfg["kind"], cfg["hidden"], seed=1, ident=cfg["ident"]))
        t_gen = time.time() - t0
        if dist is not None:
            dist.barrier()
        env = dict(os.environ, FLAPPIE_MODEL_DIR=d, FLAPPIE_HIP_DEVICE=str(local_rank), FLAPPIE_CLI_TIMING="1")
        runs, walls = {n_short: [], nfiles: []}, {}
        for _rep in range(3):                   # short and long runs in turn, the MEDIAN wall of three each: a single run's wall varies by +-5 % (process and HIP start-up, the
            for n in (n_short, nfiles):         # GPU's clock state behind the timed loop -- the first run after it is the fastest), and a minimum of each would pair unlike runs
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                r = subprocess.run([exe, "--readers", str(readers), "--shard", "%d/%d" % (rank, world), "--limit", str(n), "-o", os.path.join(d, "out.%d.fq" % rank),
                                    os.path.join(d, "reads")], env=env, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                walls.setdefault(n, []).append(round(dt, 3))
                called = [ln for ln in r.stderr.splitlines() if ln.startswith("basecalled:")]
                reads, samples, raw = (int(x) for x in (called[-1].replace(",", " ").split()[1], called[-1].split()[3], called[-1].split()[7])) if called else (0, 0, 0)
                runs[n].append((r.returncode, dt, reads, samples, raw))
        res = []
        for n in (n_short, nfiles):
            bad = [x for x in runs[n] if x[0] != 0]
            res.append(bad[0] if bad else sorted(runs[n], key=lambda x: x[1])[len(runs[n]) // 2])
        ok = gen.returncode == 0 and all(x[0] == 0 for x in res) and res[1][2] == nfiles and res[0][2] == n_short
        d_t, d_samples, d_raw = res[1][1] - res[0][1], res[1][3] - res[0][3], res[1][4] - res[0][4]
        mine = {"rank": rank, "ok": bool(ok), "marginal_s": d_t, "samples": d_samples, "raw_samples": d_raw, "long_run_s": res[1][1], "short_run_s": res[0][1],
                "Msamples_per_s": (d_raw / d_t / 1e6) if d_t > 0 else None}
        allr = [None] * world
        if dist is not None:
            dist.all_gather_object(allr, mine)
        else:
            allr = [mine]
        if rank == 0:
            tmax = max(x["marginal_s"] for x in allr)
            good = all(x["ok"] for x in allr) and tmax > 0
            out = {"value": round(sum(x["raw_samples"] for x in allr) / tmax / 1e6, 4) if good else None, "unit": "Msamples/s",
                   "per_rank": [round(x["Msamples_per_s"], 3) if x["Msamples_per_s"] else None for x in allr],
                   "max_marginal_s": round(tmax, 4), "files_per_rank": nfiles, "readers_per_rank": readers, "host_cores": ncore,
                   "fixed_cost_s": round(res[0][1] - n_short * (d_t / max(1, nfiles - n_short)), 3), "rank0_walls_s": {str(k): v for k, v in walls.items()},
                   "note": "flappie binary per rank: --shard rank/%d over one directory of %d generated single-read fast5 files (3500-5500 raw samples), "
                           "--readers %d (host cores %d / ranks %d), FASTQ out; raw samples of files [%d, %d) of each shard / the slowest rank's time between a "
                           "%d-file and a %d-file run (runs in turn, the median wall of three each); generation %.1f s (not timed)" % (world, world * nfiles, readers, ncore, world, n_short, nfiles, n_short, nfiles, t_gen)}
    finally:
        if dist is not None:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(d, ignore_errors=True)
    return out


def length_mix_leg(cfg, local_rank, nfiles=None):
    """BASELINE.json configs[2] is a DIRECTORY OF READS, and reads do not share a length: the `flappie` binary over generated single-read fast5 files of
    log-normal lengths (median 8000 samples, sigma 1, clipped to 1000 .. 200 000: fast5_tool synthln) -- one whole run, start-up included, beside `value`, never as it.
    The binary packs such a chunk several reads to a row (include/ffhip.h "packed batches"); its own account of what the batches paid for comes
    back as `padding_efficiency`.  One GPU, rank 0 only (like cpu_baseline); {"skipped": why} when the binary is missing."""
    import shutil
    import subprocess
    import tempfile
    from flappie_amd import model as M
    exe, tool = os.path.join(ROOT, "flappie_amd", "flappie"), os.path.join(ROOT, "flappie_amd", "fast5_tool")
    if cfg["kind"] != M.NET_LSTM5 or not (os.path.exists(exe) and os.path.exists(tool)):
        return {"skipped": "needs the flappie binary + fast5_tool (libhdf5 at build time) and an LSTM5 flip-flop model"}
    nfiles = nfiles or int(os.environ.get("FFHIP_BENCH_LENMIX_FILES", "49152"))
    n_short = max(256, nfiles // 3)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="ffhip_lenmix_", dir=base)
    try:
        os.mkdir(os.path.join(d, "reads"))
        t0 = time.time()
        gen = subprocess.run([tool, "synthln", os.path.join(d, "reads"), str(nfiles), "8000", "1.0", "1000", "200000", "20260930"], capture_output=True, text=True)
        M.write_mdl(os.path.join(d, "flipflop5_r941native.h"), M.synthetic_model(cfg["kind"], cfg["hidden"], seed=1, ident=cfg["ident"]))
        t_gen = time.time() - t0
        readers = max(1, min(4, effective_cpus()[0]))
        env = dict(os.environ, FLAPPIE_MODEL_DIR=d, FLAPPIE_HIP_DEVICE=str(local_rank), FLAPPIE_CLI_TIMING="1",
                   FLAPPIE_DEBUG=",".join(x for x in (os.environ.get("FLAPPIE_DEBUG", ""), "pack_log") if x))      # (pack_log: a line a packed batch with the time it was submitted)
        runs = {}
        # a short and a long run: the rate is MARGINAL (as host_fed's): start-up and the batch objects' allocation are in both.  A first short run is thrown away: what the allocation
        # of the 2 x 100 GB batch objects costs depends on what the box's memory has been through (0 .. 5 s; the first process to touch it pays most:
        # profiles/r06_length_mix.txt) -- both timed runs then start from the same state
        for k, lim in enumerate((n_short, n_short, nfiles)):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "--readers", str(readers), "--limit", str(lim), "-o", os.path.join(d, "out.fq"), os.path.join(d, "reads")], env=env, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            called = [ln for ln in r.stderr.splitlines() if ln.startswith("basecalled:")]
            pad = [ln for ln in r.stderr.splitlines() if ln.startswith("batches:")]
            if gen.returncode != 0 or r.returncode != 0 or not called:
                return {"skipped": "flappie on the mixed directory failed (rc %d): %s" % (r.returncode, r.stderr[-300:])}
            if k == 0:
                continue
            runs[lim] = (dt, int(called[-1].replace(",", " ").split()[1]), int(called[-1].split()[7]), pad[-1] if pad else "")
            phases = {}
            for ln in r.stderr.splitlines():      # the binary's own wall-clock split (FLAPPIE_CLI_TIMING) of the last run
                m = ln.rsplit(None, 2)
                if len(m) == 3 and m[2] == "s" and not ln.startswith("packed"):
                    try:
                        phases[m[0].strip()] = float(m[1])
                    except ValueError:
                        pass
            objects = [ln for ln in r.stderr.splitlines() if ln.startswith("packed batch object")]
        (t_s, _, raw_s, _), (t_l, reads, raw_l, pad) = runs[n_short], runs[nfiles]
        eff = pad.split("padding efficiency ")[1].split()[0] if "padding efficiency " in pad else None
        # The steady state by the long run's own account: the binary says when it submitted every packed batch and how many (trimmed) samples it holds; from the third batch
        # on -- both batch objects exist by then: their allocation costs 0 .. 5 s from one invocation to the next (profiles/r06_alloc_probe.txt) and makes the difference of two
        # walls scatter by tens of per cent -- the samples of batches 3 .. n - 1 over the time between the third and the last submission.  `marginal_walls` is that difference.
        import re
        sub = [(int(m.group(1)), float(m.group(2)), int(m.group(3))) for m in re.finditer(r"^packed batch (\d+) \(submitted at ([0-9.]+) s\): \d+ reads, (\d+) samples", r.stderr, re.M)]
        steady = None
        if len(sub) >= 5 and sub[-1][1] > sub[2][1]:
            steady = sum(x[2] for x in sub[2:-1]) / (sub[-1][1] - sub[2][1]) / 1e6
        marginal = (raw_l - raw_s) / (t_l - t_s) / 1e6 if t_l > t_s else None
        return {"value": round(steady if steady is not None else marginal, 3) if (steady or marginal) else None, "unit": "Msamples/s",
                "basis": "steady state: trimmed samples of the long run's packed batches 3 .. n-1 / the time between the third and the last submission (the binary's own time stamps)" if steady is not None else "marginal rate of the two walls",
                "marginal_walls": round(marginal, 3) if marginal else None, "packed_batches_long_run": len(sub), "whole_long_run": round(raw_l / t_l / 1e6, 3),
                "walls_s": {str(n_short): round(t_s, 3), str(nfiles): round(t_l, 3)}, "files": nfiles, "reads_called": reads, "raw_samples": raw_l,
                "padding_efficiency": float(eff) if eff else None, "batches": pad.split(";")[0] if pad else None, "long_run_phases_s": phases, "long_run_batch_objects": objects,
                "note": "the flappie binary over %d generated single-read fast5 files of log-normal lengths (median 8000, sigma 1, 1000 .. 200 000 samples), --readers %d, FASTQ out; `value`: see `basis`; `marginal_walls`: "
                        "raw samples of files [%d, %d) / the time between a %d-file and a %d-file run (start-up and the allocation of the ~100 GB batch objects are in both -- the latter costs 0 .. 3 s an object from one "
                        "invocation to the next, so this figure scatters: whole runs of 864.7 M samples read 83 Msamples/s in profiles/r06_length_mix.txt; `whole_long_run` has both in); padding_efficiency = samples / (rows x the batch's longest row) by the binary's own account; round 5's one-read-a-row batcher "
                        "read 13.6 Msamples/s of this mix at 0.07 (profiles/r06_length_mix.txt); generation %.1f s (not timed)" % (nfiles, readers, n_short, nfiles, n_short, nfiles, t_gen)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="workload: c2 = the headline (default); h256, c4, c5, rle = the other shapes DESIGN.md quotes")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("FFHIP_INFLIGHT", "0")) or None,
                    help="batches in flight per GPU, each on its own HIP stream (default: the workload's measured best -- 2 where one workgroup "
                         "per CU runs the layers and the other batch's convolution / decode kernels fit beside them: c2 +2.9 %%, rle +24 %%; 1 where two "
                         "workgroups per CU already fill the CUs: there a second batch costs up to 10 %%)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pair", action="store_true", help="c2 / rle: submit the batches one by one (ffhip_batch_run) instead of in pairs (ffhip_batch_run_pair)")
    ap.add_argument("--no-h2d-leg", action="store_true")
    ap.add_argument("--no-host-fed-leg", action="store_true", help="skip the per-rank run of the flappie binary over generated fast5 files")
    ap.add_argument("--no-length-mix-leg", action="store_true", help="skip the run of the flappie binary over generated fast5 files of mixed lengths (one GPU only)")
    ap.add_argument("--hidden", type=int, default=None, help="override the config's hidden size")
    ap.add_argument("--nread", type=int, default=int(os.environ.get("FFHIP_BENCH_NREAD", "0")) or None, help="override the config's reads per batch")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not os.environ.get("FFHIP_BENCH_NO_SELF_LAUNCH"):
        self_launch(args.gpus, sys.argv[1:])          # does not return
    cfg = dict(CONFIGS[args.config])
    if args.hidden:
        cfg["hidden"] = args.hidden
    if args.nread:
        cfg["nread"] = args.nread
    steps = args.steps if args.steps is not None else cfg["steps"]
    warmup = args.warmup if args.warmup is not None else cfg["warmup"]
    NREAD, NSAMPLE, H = cfg["nread"], cfg["nsample"], cfg["hidden"]

    # Only the JSON line may reach stdout: RCCL prints a version banner there at init, so fd 1 is pointed at
    # stderr for the duration of the run and the result is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # More than one rank: a host thread that waits for its GPU sleeps on the interrupt instead of spinning (FFHIP_DEBUG=blocking_sync, ffhip_engine_create:
    # hipDeviceScheduleBlockingSync).  One rank spins 2.1-2.4 CPUs away, a blocking one 1.6, at the same Msamples/s (profiles/r06_blocking_sync.txt) -- eight
    # spinning ranks would sit on the 16 CPUs a container of this pool is granted.  FFHIP_BENCH_SPIN=1 keeps HIP's default; the line says which (`host_wait`).
    if world > 1 and not os.environ.get("FFHIP_BENCH_SPIN") and "blocking_sync" not in os.environ.get("FFHIP_DEBUG", ""):
        os.environ["FFHIP_DEBUG"] = ",".join(x for x in (os.environ.get("FFHIP_DEBUG", ""), "blocking_sync") if x)
    # this rank, the flappie process of its host-fed leg and that one's reader children on the CPUs of the GPU's NUMA node (flappie_amd/shard.py;
    # FFHIP_BENCH_SYSFS: a fake sysfs tree for the CPU tests; FFHIP_BENCH_NO_NUMA_BIND=1: off)
    from flappie_amd import shard as _shard
    numa_bound = (-1, 0) if os.environ.get("FFHIP_BENCH_NO_NUMA_BIND") else _shard.bind_to_gpu_numa(local_rank, os.environ.get("FFHIP_BENCH_SYSFS", "/sys"))
    import torch
    import torch.distributed as dist
    # FFHIP_BENCH_FORCE_DIST=1: take the RCCL code path (init, barrier, MAX all-reduce) with a single rank too --
    # lets a 1-GPU box exercise exactly what the N > 1 launches run
    dist_on = world > 1 or bool(os.environ.get("FFHIP_BENCH_FORCE_DIST"))
    stub = bool(os.environ.get("FFHIP_BENCH_STUB"))          # CPU tests: gloo, no engine (tests/test_bench_launch.py)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if stub:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE %d != --gpus %d (start it as `python bench.py --gpus N`, or under torch.distributed.run with N ranks)" % (world, args.gpus))

    from flappie_amd import model as M
    mdl = M.synthetic_model(cfg["kind"], H, seed=1, ident=cfg["ident"])
    if stub:
        eng, dm = _StubEngine(), None

        class B:          # noqa: N801 -- stands for the binding module below
            Batch = _StubBatch
    else:
        from flappie_amd import binding as B
        eng = B.Engine(local_rank)
        dm = B.DeviceModel(eng, mdl)
    rng = np.random.default_rng(20260928 + rank)
    sig = rng.standard_normal((NREAD, NSAMPLE)).astype(np.float32)
    nfl = max(1, min(args.inflight if args.inflight else cfg.get("inflight", 1), 2))
    # pair: two batches go to the GPU together, their recurrent layers as ONE launch per layer (ffhip_batch_run_pair: at H = 384 a 256-read
    # batch is half of what the layer kernel's dense form carries); `nfl` then counts PAIRS in flight.  A step is still one batch.
    pair = bool(cfg.get("pair")) and not args.no_pair and not stub and nfl >= 1
    batches = [B.Batch(dm, NREAD, NSAMPLE) for _ in range(nfl * (2 if pair else 1))]
    for b in batches:
        b.set_signals(sig)               # inputs resident in HBM before the timed region

    def barrier():
        if dist_on:
            dist.barrier()
        eng.synchronize()
        if torch.cuda.is_available() and not stub:
            torch.cuda.synchronize()

    def run_steps(n, upload=False):
        pending = []
        if pair:
            for i in range(0, n, 2):
                k = (i // 2) % nfl
                b0, b1 = batches[2 * k], batches[2 * k + 1]
                if len(pending) == nfl:
                    for b in pending.pop(0):
                        b.finish()
                if upload:
                    b0.set_signals(sig)
                    b1.set_signals(sig)
                if i + 1 < n:
                    b0.run_pair(b1, 1.0, 0)
                    pending.append((b0, b1))
                else:                      # an odd step count: the last batch runs alone
                    b0.run(1.0, 0)
                    pending.append((b0,))
            for bs in pending:
                for b in bs:
                    b.finish()
            return
        for i in range(n):
            b = batches[i % nfl]
            if len(pending) == nfl:
                pending.pop(0).finish()
            if upload:
                b.set_signals(sig)       # host buffer -> HBM inside the step (the PCIe-inclusive leg)
            b.run(1.0, 0)
            pending.append(b)
        for b in pending:
            b.finish()

    run_steps(warmup)
    eng.set_profiling(True)              # HIP events on the kernels' own stream; no host synchronisation added
    barrier()
    t0 = time.perf_counter()
    run_steps(steps)
    barrier()
    dt = time.perf_counter() - t0
    prof = [b.profile() for b in batches]
    eng.set_profiling(False)
    nblock_, rnn_path_ = batches[0].nblock, batches[-1].rnn_path()      # (the batches are closed before the line is put together)
    paired_ = (not stub) and pair and batches[0].paired()
    # GRUmod at H = 256 in batches of whole 1024-read launches: the packed form of the layer kernel (ffhip_rnn_split.hip, PACK)
    packed_ = (not stub) and cfg["kind"] in (M.NET_GRUMOD5, M.NET_LSTM5) and H == 256 and dm is not None and dm.launch_reads == 1024 and NREAD % 1024 == 0

    # second leg, reported beside `value`, never as it: the same steps with the batch's signal handed over as a HOST buffer
    # every step (SURVEY.md section 8d counts "from first H2D")
    dt_h2d, steps_h2d = None, 0
    if not args.no_h2d_leg:
        steps_h2d = steps                # the same K steps as `value`
        barrier()
        t1 = time.perf_counter()
        run_steps(steps_h2d, upload=True)
        barrier()
        dt_h2d = time.perf_counter() - t1

    if dist_on:       # MAX over ranks of the timed region (flappie_amd/shard.py::max_over_ranks, inlined so that it also runs at world 1)
        dt_rank = dt
        tmax = torch.tensor([dt, dt_h2d or 0.0], dtype=torch.float64, device="cpu" if stub else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0].item())
        dt_h2d = float(tmax[1].item()) if dt_h2d is not None else None
        per_rank = [None] * world
        dist.all_gather_object(per_rank, round(steps * NREAD * NSAMPLE / dt_rank / 1e6, 4))
    else:
        per_rank = [round(steps * NREAD * NSAMPLE / dt / 1e6, 4)]

    # the engine is closed before the host-fed leg: that leg is a `flappie` process per rank on the same GPU
    for b in batches:
        b.close()
    if dm is not None:
        dm.close()
    eng.close()
    hostfed = None
    if not args.no_host_fed_leg and not stub:
        try:
            hostfed = host_fed_leg(cfg, rank, local_rank, world, dist if dist_on else None)
        except Exception as e:      # a reported leg, not the benchmark: its failure must not lose the line
            hostfed = {"skipped": "host-fed leg failed: %r" % (e,)} if rank == 0 else None
    lenmix = None
    if world == 1 and rank == 0 and not stub and not args.no_length_mix_leg and not args.no_host_fed_leg:
        try:
            lenmix = length_mix_leg(cfg, local_rank)
        except Exception as e:
            lenmix = {"skipped": "length-mix leg failed: %r" % (e,)}

    if rank == 0:
        nblock = nblock_
        value = world * steps * NREAD * NSAMPLE / dt / 1e6
        # dominant kernel: one recurrent layer.  Algorithmic work per read per block:
        # 2*H*G*H FLOP for the recurrence (G = 4 gates for LSTM, 3 for GRUmod), and the same again for the input
        # projection when the layer runs fused (projection and recurrence in one persistent launch; `inproj` then has 0 launches).
        G = 3 if cfg["kind"] == M.NET_GRUMOD5 else 4
        rec = prof[-1]["recurrent"]
        fused = prof[-1]["inproj"]["launches"] == 0
        rnn_path = rnn_path_
        # a paired launch (ffhip_batch_run_pair) carries the layer of TWO batches: its duration is recorded with both, its work is 2 x
        flop_layer = (2.0 if fused else 1.0) * 2.0 * H * G * H * NREAD * nblock * (2 if paired_ else 1)
        launches_per_layer = rec["launches"] / 5.0
        ms_layer = rec["ms"] / 5.0
        achieved = flop_layer / (ms_layer * 1e-3) / 1e12
        cell = "GRUmod" if G == 3 else "LSTM"
        if rnn_path in (3, 4):
            # fp32-grade products out of 16-bit MFMAs: each algorithmic (fp32) multiply-add is three fp16 MFMA products
            # (two-way fp16 split of both operands, the smallest cross term dropped; accuracy of an fp32 GEMM:
            # tests/test_split_numerics.py).  `achieved` counts the ALGORITHMIC fp32 FLOPs; the ceiling of this
            # formulation is the dense fp16/bf16 MFMA peak / 3.
            if rnn_path == 3 and packed_:
                kname = ("k_grumod_pack (GRUmod input projection + recurrence of one layer on fp16 MFMAs over 2-way split operands, gate-major row tiles: no empty accumulator rows; %d dependent steps)" if G == 3 else
                         "k_lstm_pack (LSTM input projection + recurrence of one layer on fp16 MFMAs over 2-way split operands, gate-major row tiles, 16 members a group; %d dependent steps)") % nblock
            elif rnn_path == 3:
                kname = "k_lstm_split%s<%d,%d> (%s input projection + recurrence of one layer on fp16 MFMAs over 2-way split operands, %d dependent steps%s)" % (
                    "_pair" if paired_ else "", 1 if G == 3 else 0, H // 128, cell, nblock, "; ONE launch for the layer of two %d-read batches" % NREAD if paired_ else "")
            else:
                kname = "k_rnn_split (%s recurrence of one layer on fp16 MFMAs over 2-way split operands, %d dependent steps; its projection GEMM k_inproj_split is a separate launch)" % (cell, nblock)
            peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS
            peak_note = ("dense fp16/bf16 MFMA peak %.0f TFLOP/s / %d products per fp32 multiply-add; the kernel issues %.1f TFLOP/s of fp16 MFMA work "
                         "= %.3f of the 16-bit MFMA peak; against the f32-input MFMA peak (%.1f) the algorithmic rate is %.3f"
                         % (PEAK_BF16_MFMA_TFLOPS, SPLIT_PRODUCTS, achieved * SPLIT_PRODUCTS, achieved * SPLIT_PRODUCTS / PEAK_BF16_MFMA_TFLOPS,
                            PEAK_F32_MFMA_TFLOPS, achieved / PEAK_F32_MFMA_TFLOPS))
            dtype = ("f32 (products as 3 fp16 MFMA terms over 2-way split operands, f32 accumulate; gate activations f32 through v_exp_f32 / v_rcp_f32 with a two-word exponent and a Newton step"
                     " -- FFHIP_FAST_GATES=0 replays the reference's exp_ps instead: profiles/r06_gates_c2.txt)")
        else:
            kname = ("k_lstm_fused (%s input projection + recurrence of one layer, %d dependent steps)" if fused
                     else "k_rnn_persist (%s recurrence of one layer, %d dependent steps)") % (cell, nblock)
            peak = PEAK_F32_MFMA_TFLOPS
            peak_note = "f32-input MFMA peak (v_mfma_f32_16x16x4_f32)"
            dtype = "f32"
        roof = {"bound": "mfma", "kernel": kname,
                "achieved": round(achieved, 3), "peak": round(peak, 2), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": measured_traffic(args.config, cfg, rnn_path, paired_)[0],
                "traffic_source": "%s: the committed PMC passes of this kernel at this shape (2 x FETCH_SIZE + WRITE_SIZE), not counters of this run"
                                  % measured_traffic(args.config, cfg, rnn_path, paired_)[1],
                "peak_note": peak_note,
                # the same algorithmic rate against the ceiling of round 1's formulation (six bf16 products per fp32 product,
                # 2500 / 6 = 416.7 TFLOP/s), which VERDICT r1 priced the kernel with (0.39 then; its target: >= 0.50)
                "frac_of_six_product_ceiling": round(achieved / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4) if rnn_path in (3, 4) else None,
                "flop_per_launch": flop_layer / launches_per_layer,
                "avg_launch_ms": round(ms_layer / launches_per_layer, 6),
                "launches_per_layer": launches_per_layer}
        if not fused and prof[-1]["inproj"]["ms"] > 0:
            ip = prof[-1]["inproj"]
            ip_flop = 2.0 * H * G * H * NREAD * nblock
            roof["inproj_gemm"] = {"achieved": round(ip_flop / (ip["ms"] / 5.0 * 1e-3) / 1e12, 3), "unit": "TFLOP/s",
                                   "frac": round(ip_flop / (ip["ms"] / 5.0 * 1e-3) / 1e12 / peak, 4), "avg_ms_per_layer": round(ip["ms"] / 5.0, 4)}
        bytes_per_block = 12 * mdl.nparam + mdl.nstate + 8
        dec_ms = prof[-1]["posterior"]["ms"] + prof[-1]["viterbi_assembly"]["ms"]
        out = {
            "metric": cfg["metric"],
            "value": round(value, 4), "unit": "Msamples/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic (seeded N(0,1) signal, seeded random-init weights of the %s architecture)" % cfg["ident"],
            "config": {"workload": cfg["label"].replace("H=%d" % CONFIGS[args.config]["hidden"], "H=%d" % H).replace("batch=%d" % CONFIGS[args.config]["nread"], "batch=%d" % NREAD), "name": args.config,
                       "reads_per_step": NREAD, "samples_per_read": NSAMPLE, "blocks_per_read": nblock,
                       "batches_in_flight": nfl * (2 if pair else 1), "paired_layer_launches": bool(paired_), "parallelism": "reads sharded by rank, no collective"},
            "roofline": roof,
            # per STEP (= one batch): a paired layer launch serves two steps, so half of its duration is this step's share
            "kernel_ms_per_step": {k: round(v["ms"] / (2.0 if (paired_ and k == "recurrent") else 1.0), 4) for k, v in prof[-1].items()},
            # what a step takes beyond its share of the layer launches: convolutions, head, decode and launch gaps that nothing hides
            "exposed_ms": round(dt / steps * 1e3 - rec["ms"] / (2.0 if paired_ else 1.0), 4),
            "kernel_ms_note": ("one batch in flight: the kernels of a step run back to back" if nfl == 1 else
                               "two batches (or pairs) in flight: between two batches' layer launches the head / decode kernels of the one run BESIDE the "
                               "convolutions of the next (ordered by the engine: FFHIP_DEBUG=front_order=...), so their durations here are stretched by each other, "
                               "overlap, and do not add up to ms_per_step; `--inflight 1` gives the serial breakdown"),
            # decode side (posterior + Viterbi + assembly + trace): algorithmic bytes per block (SURVEY.md section 8d:
            # 4P read + nstate traceback + 8 path/qpath, plus 4P read + 4P write for the posterior) against HBM peak.
            # At 256 reads these kernels are latency-bound chains, not bandwidth-bound.  For the 8- and 10-state models the
            # "posterior" time is the launch of k_crf_fb + k_post_fb: the partition function's chain, the normalisation and
            # the posterior together (they share the chains), so this rate is a lower bound of the decode side's own.
            "decode_hbm": {"achieved": round(float(NREAD) * nblock * bytes_per_block / (dec_ms * 1e-3) / 1e9, 2) if dec_ms > 0 else None,
                           "peak": 8000.0, "unit": "GB/s", "bytes_per_block": bytes_per_block, "ms": round(dec_ms, 4)},
        }
        out["per_rank_Msamples_per_s"] = per_rank
        out["host_wait"] = "blocking (hipDeviceScheduleBlockingSync)" if "blocking_sync" in os.environ.get("FFHIP_DEBUG", "") else "spin (HIP's default)"
        out["host_binding"] = {"rank0_numa_node": numa_bound[0], "rank0_cpus": numa_bound[1] or len(os.sched_getaffinity(0)),
                               "note": "every rank binds itself (and the flappie process + reader children of its host-fed leg) to the CPUs of its GPU's NUMA node "
                                       "that it may use; node -1 = not known or outside this container's CPU set: left alone"}
        out["max_over_ranks_s"] = round(dt, 6)
        if hostfed is not None:
            out["host_fed"] = hostfed
        if lenmix is not None:
            out["length_mix"] = lenmix
        if dt_h2d is not None:
            out["h2d_inclusive"] = {"value": round(world * steps_h2d * NREAD * NSAMPLE / dt_h2d / 1e6, 4), "unit": "Msamples/s",
                                    "ms_per_step": round(dt_h2d / steps_h2d * 1e3, 4), "steps": steps_h2d,
                                    "note": "the same step with the batch's %.1f MB of signal copied from a host buffer inside it (never `value`)" % (NREAD * NSAMPLE * 4 / 1e6)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
