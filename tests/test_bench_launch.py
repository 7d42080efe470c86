"""bench.py's launch / timing / reduction plumbing: `python bench.py --gpus N` started by hand starts its own N
ranks under torch.distributed.run; every rank times the same K steps between barriers, the MAX over ranks divides the whole job's
samples.  On CPU the engine is a stub (FFHIP_BENCH_STUB=1: gloo instead of RCCL, a sleep instead of a batch) -- this tests the plumbing,
not the kernels; the GPU test runs the real line at world size 1 through the same distributed code path, host-fed leg included."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    return r


def _line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 8])
def test_self_launch_with_a_stub_engine(n):
    env = {"FFHIP_BENCH_STUB": "1"}
    if n == 1:
        env["FFHIP_BENCH_FORCE_DIST"] = "1"          # the distributed code path (init, barrier, MAX all-reduce, gather) at world size 1
    r = _run(["--gpus", str(n), "--steps", "6", "--warmup", "1", "--no-cpu-baseline"], env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r)
    assert d["n_gpus"] == n and d["steps"] == 6 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert len(d["per_rank_Msamples_per_s"]) == n
    samples = n * 6 * d["config"]["reads_per_step"] * d["config"]["samples_per_read"]
    assert abs(d["value"] - samples / d["max_over_ranks_s"] / 1e6) <= 1e-3 * d["value"]          # whole job / slowest rank
    assert d["value"] <= sum(d["per_rank_Msamples_per_s"]) * (1 + 1e-6)                            # MAX over ranks, not a sum of rates
    assert abs(d["ms_per_step"] - d["max_over_ranks_s"] / 6 * 1e3) < 1e-3
    # more than one rank: the ranks' host threads wait for their GPUs blocking, not spinning (profiles/r06_blocking_sync.txt), and the line says so
    assert d["host_wait"].startswith("blocking" if n > 1 else "spin")


def test_world_size_mismatch_is_an_error_message_not_an_assertion():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"FFHIP_BENCH_STUB": "1", "FFHIP_BENCH_NO_SELF_LAUNCH": "1"})
    assert r.returncode != 0 and "WORLD_SIZE 1 != --gpus 2" in r.stderr and "Traceback" not in r.stderr


@pytest.mark.gpu
def test_real_line_through_the_distributed_path_with_host_fed_leg():
    r = _run(["--steps", "4", "--warmup", "1", "--no-cpu-baseline"], {"FFHIP_BENCH_FORCE_DIST": "1", "FFHIP_BENCH_HOSTFED_FILES": "1536", "FFHIP_BENCH_LENMIX_FILES": "1024"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r)
    assert d["n_gpus"] == 1 and d["value"] > 1.0 and "roofline" in d and len(d["per_rank_Msamples_per_s"]) == 1
    hf = d.get("host_fed")
    assert hf is not None
    if "skipped" not in hf:          # (needs libhdf5 at build time)
        assert hf["value"] and hf["value"] > 0.5 and len(hf["per_rank"]) == 1 and hf["files_per_rank"] == 1536
    lm = d.get("length_mix")          # the binary on a directory of mixed read lengths (packed batches), one whole run
    assert lm is not None
    if "skipped" not in lm:
        assert lm["files"] == 1024 and lm["reads_called"] == 1024 and lm["whole_long_run"] > 0.5 and lm["padding_efficiency"] > 0.5


def _fake_sysfs(tmp_path, nodes):
    """a sysfs tree with one AMD render node per entry of `nodes` (its NUMA node), and two NUMA nodes that split this process's CPUs"""
    cpus = sorted(os.sched_getaffinity(0))
    half = max(1, len(cpus) // 2)
    lists = {0: cpus[:half], 1: cpus[half:] or cpus[:half]}
    for k, node in enumerate(nodes):
        d = tmp_path / "class" / "drm" / ("renderD%d" % (128 + k)) / "device"
        d.mkdir(parents=True)
        (d / "vendor").write_text("0x1002\n")
        (d / "numa_node").write_text("%d\n" % node)
    other = tmp_path / "class" / "drm" / "renderD200" / "device"
    other.mkdir(parents=True)
    (other / "vendor").write_text("0x10de\n")
    (other / "numa_node").write_text("0\n")
    for node, cl in lists.items():
        d = tmp_path / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cl) + "\n")
    return lists


def test_numa_binding_of_a_rank(tmp_path):
    """a rank binds itself (and what it starts) to the CPUs of its GPU's NUMA node -- flappie_amd/shard.py for bench.py's ranks, the
    same walk over sysfs in the flappie binary (before it forks its readers).  Against a fake sysfs tree."""
    sys.path.insert(0, ROOT)
    from flappie_amd import shard
    assert shard.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    lists = _fake_sysfs(tmp_path, [0, 1, 1, -1])
    assert shard.numa_cpus_of_gpu(0, str(tmp_path)) == (0, lists[0])
    assert shard.numa_cpus_of_gpu(2, str(tmp_path)) == (1, lists[1])
    assert shard.numa_cpus_of_gpu(3, str(tmp_path)) == (-1, []) and shard.numa_cpus_of_gpu(7, str(tmp_path)) == (-1, [])
    code = ("import os, sys; sys.path.insert(0, %r); from flappie_amd import shard; r = shard.bind_to_gpu_numa(1, %r); "
            "print(r[0], r[1], sorted(os.sched_getaffinity(0)))" % (ROOT, str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.split(None, 2)
    assert int(out[0]) == 1 and int(out[1]) == len(lists[1]) and eval(out[2]) == lists[1]
    exe = os.path.join(ROOT, "flappie_amd", "flappie")
    if os.path.exists(exe):
        reads = tmp_path / "reads"
        reads.mkdir()
        (reads / "a.fast5").write_text("")
        env = dict(os.environ, FLAPPIE_DEBUG="list_only,sysfs_root=%s" % tmp_path, FLAPPIE_HIP_DEVICE="2", FLAPPIE_CLI_TIMING="1")
        r = subprocess.run([exe, str(reads)], env=env, capture_output=True, text=True)
        assert r.returncode == 0 and "bound to %d CPUs of NUMA node 1 (GPU 2)" % len(lists[1]) in r.stderr, r.stderr
        r = subprocess.run([exe, str(reads)], env=dict(env, FLAPPIE_DEBUG="list_only,no_numa_bind,sysfs_root=%s" % tmp_path), capture_output=True, text=True)
        assert r.returncode == 0 and "bound to" not in r.stderr


def test_numa_binding_goes_through_the_visible_devices_lists(tmp_path):
    """device n of a process is the n-th render node only when no *_VISIBLE_DEVICES variable re-maps it; the index is taken through the lists
    (HIP's, or its synonym CUDA's, picks from what ROCR's leaves) and a UUID entry or a short list leaves the process unbound."""
    sys.path.insert(0, ROOT)
    from flappie_amd import shard
    f = shard.physical_gpu_index
    assert f(3, {}) == 3
    assert f(0, {"HIP_VISIBLE_DEVICES": "5"}) == 5 and f(1, {"HIP_VISIBLE_DEVICES": "5, 2"}) == 2
    assert f(1, {"CUDA_VISIBLE_DEVICES": "4,6"}) == 6 and f(1, {"CUDA_VISIBLE_DEVICES": "4,6", "HIP_VISIBLE_DEVICES": "7,1"}) == 1
    assert f(1, {"HIP_VISIBLE_DEVICES": "2,0", "ROCR_VISIBLE_DEVICES": "4,5,6"}) == 4
    assert f(1, {"HIP_VISIBLE_DEVICES": "5"}) == -1 and f(0, {"ROCR_VISIBLE_DEVICES": "GPU-abcdef"}) == -1
    exe = os.path.join(ROOT, "flappie_amd", "flappie")
    if os.path.exists(exe):
        lists = _fake_sysfs(tmp_path, [0, 0, 1, 1])
        reads = tmp_path / "reads"
        reads.mkdir()
        (reads / "a.fast5").write_text("")
        base = {k: v for k, v in os.environ.items() if not k.endswith("_VISIBLE_DEVICES")}
        env = dict(base, FLAPPIE_DEBUG="list_only,sysfs_root=%s" % tmp_path, FLAPPIE_HIP_DEVICE="0", FLAPPIE_CLI_TIMING="1")
        r = subprocess.run([exe, str(reads)], env=dict(env, HIP_VISIBLE_DEVICES="3"), capture_output=True, text=True)
        assert r.returncode == 0 and "CPUs of NUMA node 1" in r.stderr, r.stderr
        r = subprocess.run([exe, str(reads)], env=dict(env, HIP_VISIBLE_DEVICES="1,0", ROCR_VISIBLE_DEVICES="2,1"), capture_output=True, text=True)
        assert r.returncode == 0 and "CPUs of NUMA node 0" in r.stderr, r.stderr
        r = subprocess.run([exe, str(reads)], env=dict(env, ROCR_VISIBLE_DEVICES="GPU-0123abcd"), capture_output=True, text=True)
        assert r.returncode == 0 and "bound to" not in r.stderr, r.stderr


def test_bench_line_reports_the_binding(tmp_path):
    _fake_sysfs(tmp_path, [1])
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"], {"FFHIP_BENCH_STUB": "1", "FFHIP_BENCH_SYSFS": str(tmp_path)})
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r)
    assert d["host_binding"]["rank0_numa_node"] == 1 and d["host_binding"]["rank0_cpus"] >= 1
