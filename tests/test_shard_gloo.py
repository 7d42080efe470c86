"""The N>1 path on CPU: two gloo ranks shard reads, "basecall" their shards (the oracle stands in for
the GPU here -- this tests the sharding, gathering and timing plumbing, not the kernels) and rank 0
ends up with exactly the single-process result in input order."""
import os
import socket

import numpy as np
import pytest

from flappie_amd import model as M
from flappie_amd import shard


def test_partition_properties():
    rng = np.random.default_rng(0)
    n = rng.integers(1000, 100000, size=37).tolist()
    for world in (1, 2, 4, 8):
        parts = shard.partition_reads(n, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(37))
        loads = [sum(n[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(n)          # LPT bound
    assert shard.partition_reads([], 2) == [[], []]
    b = shard.bucket_by_length([0, 1, 2, 3, 4], [500, 700, 500, 500, 700], max_batch=2)
    assert b == [[0, 2], [3], [1, 4]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, signals, out):
    import torch.distributed as dist
    from oracle import ffo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mdl = M.synthetic_model(M.NET_LSTM5, 16, seed=2)
    om = ffo.OracleModel(mdl)
    lens = [len(s) for s in signals]
    mine = shard.partition_reads(lens, world)[rank]
    local = {}
    for batch in shard.bucket_by_length(mine, lens):
        assert len({lens[i] for i in batch}) == 1
        for i in batch:
            r = om.basecall(signals[i], want_trans=False)
            local[i] = (r["basecall"], r["quality"])
    dist.barrier()
    t = shard.max_over_ranks(1.0 + rank)
    calls = shard.gather_calls(local, len(signals))
    if rank == 0:
        out.put((calls, t))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import ffo
    rng = np.random.default_rng(4)
    signals = [rng.standard_normal(n).astype(np.float32) for n in (300, 400, 300, 500, 400, 300, 350)]
    mdl = M.synthetic_model(M.NET_LSTM5, 16, seed=2)
    om = ffo.OracleModel(mdl)
    want = []
    for s in signals:
        r = om.basecall(s, want_trans=False)
        want.append((r["basecall"], r["quality"]))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, signals, q)) for r in range(2)]
    for p in procs:
        p.start()
    calls, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert calls == want
    assert t == 2.0           # MAX over ranks of (1.0, 2.0)
