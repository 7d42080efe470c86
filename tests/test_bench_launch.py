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


@pytest.mark.parametrize("n", [1, 2])
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


def test_world_size_mismatch_is_an_error_message_not_an_assertion():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"FFHIP_BENCH_STUB": "1", "FFHIP_BENCH_NO_SELF_LAUNCH": "1"})
    assert r.returncode != 0 and "WORLD_SIZE 1 != --gpus 2" in r.stderr and "Traceback" not in r.stderr


@pytest.mark.gpu
def test_real_line_through_the_distributed_path_with_host_fed_leg():
    r = _run(["--steps", "4", "--warmup", "1", "--no-cpu-baseline"], {"FFHIP_BENCH_FORCE_DIST": "1", "FFHIP_BENCH_HOSTFED_FILES": "1536"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r)
    assert d["n_gpus"] == 1 and d["value"] > 1.0 and "roofline" in d and len(d["per_rank_Msamples_per_s"]) == 1
    hf = d.get("host_fed")
    assert hf is not None
    if "skipped" not in hf:          # (needs libhdf5 at build time)
        assert hf["value"] and hf["value"] > 0.5 and len(hf["per_rank"]) == 1 and hf["files_per_rank"] == 1536
