#!/usr/bin/env python3
"""Rehearsal of EIGHT ranks' host load on a box with ONE GPU (reference behaviour README.md:81-83,
flappie.c:334-385: one process per shard of the file list, no traffic between them).

Eight `flappie` processes run at the same time over the eight shards of one directory of single-read fast5 files, each with the readers a
rank gets on a node (host cores / 8, at most 12): process 0 drives the real GPU; processes 1-7 run with FFHIP_DEBUG_HOST_REHEARSAL_MSPS set
(flappie_amd/csrc/ffhip_engine.hip) -- no network is evaluated there, every batch "takes" its samples / that rate on an emulated GPU and
returns placeholder calls, while the fast5 readers, the signal preparation with its uploads, the batch buffers, the result copies and
the FASTQ writer are the real thing.  The host then carries what eight ranks put on it.  Reported per process: the MARGINAL host-fed rate
(a long run minus a short run, both with all eight running), next to the rate of process 0 running ALONE; host CPU use during the long
run; the same with the files on /dev/shm and on disk.   Run on the GPU box:  python tools/host_scaling.py [--hidden 384] [--files 12288]"""
import argparse
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flappie_amd import model as M  # noqa: E402

EXE, TOOL = os.path.join(ROOT, "flappie_amd", "flappie"), os.path.join(ROOT, "flappie_amd", "fast5_tool")
HOOKS = os.path.join(ROOT, "tools", "test_hooks")          # `make -C flappie_amd/csrc hooks` (__graft_entry__.build() does it)
if not os.path.exists(os.path.join(HOOKS, "libffhip.so")):
    sys.exit("tools/test_hooks/libffhip.so is missing: make -C flappie_amd/csrc hooks")
NSHARD = 8
LAST_THROTTLE = (0, 0.0)
NOGPU = True          # emulated processes keep off the physical GPU altogether (--emu-on-gpu: their signal preparation and copies run on it)


def cpu_times():
    with open("/proc/stat") as fh:
        v = [int(x) for x in fh.readline().split()[1:]]
    return sum(v), v[3] + v[4]          # total, idle + iowait


def throttled():
    """(periods in which this container's CPU quota ran out, seconds its threads then stood still) so far -- cgroup v2 cpu.stat, (0, 0.0) where there is none"""
    try:
        kv = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0)) / 1e6
    except (OSError, ValueError):
        return 0, 0.0


def run_set(d, n, readers, procs, gpu_rate, by_size, real=(0,)):
    """start one flappie per shard in `procs` at once; returns {shard: (wall, reads, raw samples, fallbacks)} and the host's CPU use"""
    env0 = dict(os.environ, FLAPPIE_MODEL_DIR=d, FLAPPIE_HIP_DEVICE="0", FLAPPIE_CLI_TIMING="1")
    c0 = cpu_times()
    th0 = throttled()
    ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    ps = {}
    t0 = time.perf_counter()
    for g in procs:
        env = dict(env0)
        if g not in real:
            env["FFHIP_DEBUG_HOST_REHEARSAL_MSPS"] = str(gpu_rate)
            env["LD_LIBRARY_PATH"] = HOOKS + os.pathsep + env.get("LD_LIBRARY_PATH", "")      # the -DFFHIP_TEST_HOOKS library: the release one has no such hook
            if NOGPU:
                env["FFHIP_DEBUG_HOST_REHEARSAL_NOGPU"] = "1"
            # the rehearsal hook knows the one-read-a-row batches only (ffhip_batch_set_prepared_packed refuses under it, and a refused packed batch makes the
            # binary create its packed objects on the ONE physical GPU first: eight processes queue there, round 6's first rehearsal)
            env["FLAPPIE_DEBUG"] = ",".join(x for x in (env.get("FLAPPIE_DEBUG", ""), "no_pack") if x)
        cmd = [EXE, "--readers", str(readers), "--shard", "%d/%d" % (g, NSHARD)] + (["--shard-by-size"] if by_size else []) + ["--limit", str(n), "-o", os.path.join(d, "out.%d.fq" % g), os.path.join(d, "reads")]
        ps[g] = (subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), time.perf_counter())
    out = {}
    for g, (p, ts) in ps.items():
        _, err = p.communicate()
        dt = time.perf_counter() - ts
        called = [ln for ln in err.splitlines() if ln.startswith("basecalled:")]
        reads, raw = (int(called[-1].split()[1]), int(called[-1].split()[7])) if called else (0, 0)
        out[g] = (dt, reads, raw, err.count("falling back"), p.returncode, [ln for ln in err.splitlines() if ln.endswith(" s") and not ln.startswith("ffhip")])
    c1 = cpu_times()
    ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    busy = 1.0 - (c1[1] - c0[1]) / max(1, c1[0] - c0[0])
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)       # this set's own processes (readers included), not the host's other tenants
    th1 = throttled()
    global LAST_THROTTLE
    LAST_THROTTLE = (th1[0] - th0[0], th1[1] - th0[1])
    return out, busy, time.perf_counter() - t0, cpu_s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=384)
    ap.add_argument("--files", type=int, default=32768, help="files per shard in the long run (the short run takes a quarter)")
    ap.add_argument("--gpu-rate", type=float, default=None, help="Msamples/s of the emulated GPUs (default: bench.py's value of the shape: 104 at H = 384, 203 at H = 256)")
    ap.add_argument("--where", default="shm,disk")
    ap.add_argument("--readers", type=int, default=0, help="reader processes per flappie process (default: what bench.py gives a rank: granted CPUs / 8 - 2, 1..12)")
    ap.add_argument("--phases", action="store_true", help="print the binary's own phase times of the last process of each leg")
    ap.add_argument("--legs", default="real1,emu1,emu8,mixed")
    ap.add_argument("--emu-on-gpu", action="store_true", help="the emulated processes run their signal preparation, uploads and result copies on the one physical GPU "
                    "(eight contexts on one device: measures that device's scheduler more than the host)")
    a = ap.parse_args()
    global NOGPU
    NOGPU = not a.emu_on_gpu
    rate = a.gpu_rate or {384: 104.0, 256: 203.0}.get(a.hidden, 100.0)
    import bench
    ncore, nhw, why = bench.effective_cpus()
    readers = a.readers or max(1, min(4, ncore // NSHARD))       # (bench.py's rule: what a rank of an eight-rank job gets)
    n_short = max(512, a.files // 4)
    print("# tools/host_scaling.py --hidden %d --files %d%s: %d CPUs granted to this container, %d reader processes per flappie process, emulated GPUs at %.0f Msamples/s (%s)"
          % (a.hidden, a.files, " --emu-on-gpu" if a.emu_on_gpu else "", ncore, readers, rate,
             "their signal preparation and copies on the physical GPU" if a.emu_on_gpu else "emulated processes never touch the physical GPU in steady state"))
    for where in a.where.split(","):
        base = "/dev/shm" if where == "shm" else tempfile.gettempdir()
        d = tempfile.mkdtemp(prefix="ffhip_hostscale_", dir=base)
        try:
            os.mkdir(os.path.join(d, "reads"))
            t0 = time.time()
            gens = [subprocess.Popen([TOOL, "synth", os.path.join(d, "reads"), str(a.files), "3500", "5500", "20260928", str(g), str(NSHARD)], stdout=subprocess.DEVNULL) for g in range(NSHARD)]
            assert all(p.wait() == 0 for p in gens)
            M.write_mdl(os.path.join(d, "flipflop5_r941native.h"), M.synthetic_model(M.NET_LSTM5, a.hidden, seed=1, ident="r941native"))
            nbytes = sum(os.path.getsize(os.path.join(d, "reads", f)) for f in os.listdir(os.path.join(d, "reads")))
            print("\n## files on %s (%s): %d files, %.2f GB, generated in %.0f s" % (where, base, NSHARD * a.files, nbytes / 1e9, time.time() - t0))
            res, cost = {}, {}
            # real1: process 0 on the real GPU, alone.  emu1: one process on an emulated GPU, alone.  emu8: EIGHT processes on emulated GPUs -- the host
            # carries eight pipelines and the one physical GPU only their signal preparation and copies.  mixed: process 0 on the real GPU beside seven
            # emulated ones (there the emulated processes' small GPU operations queue behind process 0's chip-filling layer launches -- an artefact of
            # sharing ONE GPU that a node with eight does not have -- so their own rates say nothing; process 0's rate is the figure of that row).
            for label, procs, real in (("real1", [0], (0,)), ("emu1", [1], ()), ("emu2", [0, 1], ()), ("emu4", [0, 1, 2, 3], ()), ("emu8", list(range(NSHARD)), ()), ("mixed", list(range(NSHARD)), (0,))):
                if label not in a.legs.split(","):
                    continue
                runs = []
                for n in (n_short, a.files):
                    best = None
                    for _rep in range(2):
                        r = run_set(d, n, readers, procs, rate, False, real)
                        if best is None or r[2] < best[2]:      # the faster of two
                            best = r
                    runs.append(best)
                res[label] = runs
                (s_out, _, _, s_cpu), (l_out, busy, wall, l_cpu) = runs
                d_raw = sum(l_out[g][2] - s_out[g][2] for g in procs)
                for g in procs:
                    dt, raw = l_out[g][0] - s_out[g][0], l_out[g][2] - s_out[g][2]
                    ok = l_out[g][4] == 0 and l_out[g][1] == a.files
                    print("%-6s process %d (%s): marginal %.1f Msamples/s (%d raw samples in %.3f s; long run %.2f s, short %.2f s)%s%s"
                          % (label, g, "real GPU" if g in real else "emulated GPU", raw / dt / 1e6 if dt > 0 else float("nan"), raw, dt, l_out[g][0], s_out[g][0],
                             "" if ok else "  ** run failed or incomplete **", ("  [%d fall-backs to the step kernels]" % l_out[g][3]) if l_out[g][3] else ""))
                print("%-6s the container's CPU quota ran out in %d scheduler periods of the last run (its threads stood still for %.2f thread-seconds in all)" % ((label,) + LAST_THROTTLE))
                if a.phases:
                    g = procs[0] if real else procs[-1]
                    print("       phases of process %d in the long run (FLAPPIE_CLI_TIMING): %s" % (g, "; ".join(" ".join(ln.split()) for ln in l_out[g][5])))
                print("%-6s CPU time of these processes (readers included): %.2f s in the long run over %.2f s wall = %.1f CPUs; marginal %.3f CPU-s per million raw samples  "
                      "[whole host, other tenants included: %.1f %% of %d hardware threads busy]" % (label, l_cpu, wall, l_cpu / wall, (l_cpu - s_cpu) / max(1.0, d_raw / 1e6), 100 * busy, nhw))
                cost[label] = (l_cpu - s_cpu) / max(1.0, d_raw / 1e6)
            def marg(label, g):
                (s_out, _, _, _), (l_out, _, _, _) = res[label]
                return (l_out[g][2] - s_out[g][2]) / (l_out[g][0] - s_out[g][0]) / 1e6
            for k in (2, 4):
                if "emu%d" % k in res:
                    ek = [marg("emu%d" % k, g) for g in range(k)]
                    print("=> %d processes on emulated GPUs at once: %.1f ... %.1f each, %.0f in all" % (k, min(ek), max(ek), sum(ek)))
            if not all(x in res for x in ("real1", "emu1", "emu8", "mixed")):
                continue
            e1, e8 = marg("emu1", 1), [marg("emu8", g) for g in range(NSHARD)]
            print("=> host side alone (one process, emulated GPU): %.1f Msamples/s; eight at once: %.1f ... %.1f each (slowest %.2f of alone), %.0f in all"
                  % (e1, min(e8), max(e8), min(e8) / e1, sum(e8)))
            c = cost.get("emu1", cost.get("real1"))
            print("=> host CPU per million raw samples: %.3f CPU-s (one process) -- a rank at %.0f Msamples/s needs %.1f CPUs, eight need %.0f; this container is granted %d (%s)"
                  % (c, rate, c * rate, 8 * c * rate, ncore, why))
            print("=> process 0 on the real GPU: alone %.1f Msamples/s; beside seven emulated neighbours %.1f (%.2f of alone)" % (marg("real1", 0), marg("mixed", 0), marg("mixed", 0) / marg("real1", 0)))
            # the same list dealt by size: the spread of the shards' sample sums (no run needed for that)
            for flag in ([], ["--shard-by-size"]):
                sums = []
                for g in range(NSHARD):
                    r = subprocess.run([EXE, "--shard", "%d/%d" % (g, NSHARD)] + flag + [os.path.join(d, "reads")], env=dict(os.environ, FLAPPIE_DEBUG="list_only"), capture_output=True, text=True)
                    sums.append(sum(os.path.getsize(p) for p in r.stdout.split()))
                print("shard byte sums %s: min %.4f GB, max %.4f GB (max / min %.4f)" % ("by size " if flag else "by index", min(sums) / 1e9, max(sums) / 1e9, max(sums) / min(sums)))
        finally:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
