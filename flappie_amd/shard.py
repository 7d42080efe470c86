"""Multi-GPU sharding of reads.

Reads are independent units (the reference loops over files, flappie.c:364-385, and its README runs one
process per core with GNU parallel): the path shards by read, one process per GPU, with NO data-path
collective.  The only inter-rank traffic is host-side: a barrier around timed regions, a MAX reduction
of wall time, and gathering the called strings to rank 0 in input order.  All three run on whatever
torch.distributed backend the job was started with (RCCL on the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple


def partition_reads(nsamples: Sequence[int], world: int) -> List[List[int]]:
    """Deal reads to `world` ranks, balancing the sum of samples (longest-processing-time greedy).
    Deterministic: ties go to the lowest rank; every read appears exactly once."""
    order = sorted(range(len(nsamples)), key=lambda i: (-int(nsamples[i]), i))
    load = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(nsamples[i])
    for s in shards:
        s.sort()
    return shards


def bucket_by_length(indices: Sequence[int], nsamples: Sequence[int], max_batch: int = 256) -> List[List[int]]:
    """Group a rank's reads into batches of equal trimmed length (reads are never padded or split in
    time: the CRF normaliser and the recurrent state are whole-read quantities)."""
    by_len: Dict[int, List[int]] = {}
    for i in indices:
        by_len.setdefault(int(nsamples[i]), []).append(i)
    batches = []
    for n in sorted(by_len):
        idx = by_len[n]
        for k in range(0, len(idx), max_batch):
            batches.append(idx[k:k + max_batch])
    return batches


def gather_calls(local: Dict[int, Tuple[str, str]], nread: int):
    """Collect {read index: (bases, qualities)} from every rank; rank 0 gets the list in input order."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local[i] for i in range(nread)]
    parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local, parts, dst=0)
    if dist.get_rank() != 0:
        return None
    merged: Dict[int, Tuple[str, str]] = {}
    for p in parts:
        merged.update(p)
    return [merged[i] for i in range(nread)]


def max_over_ranks(seconds: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
