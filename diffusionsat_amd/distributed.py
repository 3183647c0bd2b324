"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

The path shards embarrassingly: graphs of a batch are independent (block-diagonal adjacency;
every cross-row operation -- PairNorm means, map selection, SAT check -- is per graph), so
whole formulas / whole batches go to ranks and there is NO collective inside the sampling
loop.  The only exchanges are the weight broadcast at start-up and the merge of the result
dictionaries at the end.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist

from .batch import Instance, instance_size, sat_node_count


def broadcast_weight_blob(blob: torch.Tensor, src: int = 0, force: bool = False) -> torch.Tensor:
    """Rank ``src`` holds the flat fp32 weight blob (weights.to_blob); afterwards every rank does.
    One 3.4 MB broadcast: latency-bound, outside any timed region.  ``force``: issue the collective even in a
    world of one rank (the one-GPU test that proves RCCL loads and runs on the box, tests/test_gpu_rccl.py)."""
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size() > 1):
        dist.broadcast(blob, src=src)
    return blob


def shard_instances(instances: Sequence[Instance], rank: int, world_size: int) -> List[Instance]:
    """Whole formulas to ranks, largest first onto the least-loaded rank (node count
    ``2n+m``, data/dimac.py:172-174), deterministic on every rank."""
    order = sorted(range(len(instances)), key=lambda i: (-sat_node_count(*instance_size(instances[i])), i))
    load = [0] * world_size
    mine: List[int] = []
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        load[r] += sat_node_count(*instance_size(instances[i]))
        if r == rank:
            mine.append(i)
    return [instances[i] for i in sorted(mine)]


def merge_counts(local: Dict[int, int]) -> Dict[int, int]:
    """Union of the per-rank ``{assignment int: count}`` dictionaries of ``samples()`` on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local)
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, {str(k): v for k, v in local.items()})   # ints can exceed 64 bits
    out: Dict[int, int] = {}
    for d in gathered:
        for k, v in d.items():
            out[int(k)] = out.get(int(k), 0) + v
    return out


def split_sample_budget(n_samples: int, rank: int, world_size: int) -> int:
    """How many of ``n_samples`` satisfying samples this rank should produce."""
    base, rem = divmod(int(n_samples), world_size)
    return base + (1 if rank < rem else 0)
