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


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)"""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def numa_cpus_of_gpu(index: int, sysfs: str = "/sys") -> Tuple[int, List[int]]:
    """(NUMA node, its CPUs) of the index-th AMD GPU as the kernel lists them (render nodes of vendor 0x1002 in minor order -- the order HIP numbers
    the devices in when no *_VISIBLE_DEVICES variable re-maps them); (-1, []) when the node is not known.  A rank binds itself, its reader
    children and with them its pinned staging to these CPUs before it starts (`bind_to_gpu_numa`): eight ranks of a node otherwise run their
    readers wherever the scheduler puts them, and half of every rank's host traffic crosses the socket link."""
    import glob
    import os
    nodes = []
    for path in glob.glob(os.path.join(sysfs, "class", "drm", "renderD*")):
        try:
            with open(os.path.join(path, "device", "vendor")) as fh:
                if fh.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(path, "device", "numa_node")) as fh:
                nodes.append((int(os.path.basename(path)[len("renderD"):]), int(fh.read().strip())))
        except (OSError, ValueError):
            continue
    nodes.sort()
    if not (0 <= index < len(nodes)) or nodes[index][1] < 0:
        return -1, []
    node = nodes[index][1]
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as fh:
            return node, parse_cpulist(fh.read())
    except (OSError, ValueError):
        return node, []


def physical_gpu_index(index: int, environ=None) -> int:
    """HIP device `index` of this process -> position among the node's GPUs as the kernel lists them, through the re-mapping variables:
    HIP_VISIBLE_DEVICES (or its synonym CUDA_VISIBLE_DEVICES) picks from what ROCR_VISIBLE_DEVICES leaves.  -1 when an entry is not a plain number
    (a UUID) or the index is beyond the list: the caller then leaves the process unbound -- a wrong socket is worse than none."""
    import os
    env = os.environ if environ is None else environ
    for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        text = env.get(name, "")
        if name == "CUDA_VISIBLE_DEVICES" and env.get("HIP_VISIBLE_DEVICES", ""):
            continue                                  # (HIP reads its own variable first)
        if not text.strip():
            continue
        items = [t.strip() for t in text.split(",")]
        if not (0 <= index < len(items)) or not items[index].isdigit():
            return -1
        index = int(items[index])
    return index


def bind_to_gpu_numa(index: int, sysfs: str = "/sys") -> Tuple[int, int]:
    """restrict this process (and what it starts) to the CPUs it may use that sit on the GPU's NUMA node; (node, CPUs bound to), (-1, 0) = left alone
    (node unknown, or none of its CPUs is in this process's set -- a container's quota may lie elsewhere)"""
    import os
    index = physical_gpu_index(index)
    if index < 0:
        return -1, 0
    node, cpus = numa_cpus_of_gpu(index, sysfs)
    if node < 0 or not cpus or not hasattr(os, "sched_setaffinity"):
        return -1, 0
    mine = os.sched_getaffinity(0) & set(cpus)
    if not mine:
        return -1, 0
    os.sched_setaffinity(0, mine)
    return node, len(mine)
