"""Sharding of independent environments over the GPUs of one node.

Environments never interact, so rank r of N simply owns the global environments
``[r * E / N, (r + 1) * E / N)`` (``env_id_offset`` tells the kernels where the shard starts: it
keys the layout choice and the on-device agent order, so a sharded run is identical to a
one-GPU run of the same global batch).  No collective sits on the step path; the only
exchange is a sum all-reduce of the 64-byte episode-statistics vector (RCCL when the tensor is
on the GPU, gloo on the CPU).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def rank_world() -> Tuple[int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard: returns (first global environment, number of environments)."""
    base, extra = divmod(total_envs, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def init(backend: str | None = None) -> Tuple[int, int]:
    """Join the process group set up by torch.distributed.run (no-op for a single process)."""
    rank, world = rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group(backend, **kw)
    return rank, world


def allreduce_stats(stats: torch.Tensor) -> torch.Tensor:
    """Combine the per-shard int64[8] statistics vectors over all ranks (in place): every entry is summed
    except [5], the number of step launches, which every rank counts for itself and of which the maximum
    is kept (a sum would read as world_size times the steps taken)."""
    if dist.is_available() and dist.is_initialized():  # (a group of one rank still goes through the backend)
        launches = stats[5:6].clone()
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(launches, op=dist.ReduceOp.MAX)
        stats[5:6] = launches
    return stats


def make_sharded(spec: dict, total_envs: int, **kw):
    """This rank's ``BatchedMultiGrid`` shard of a global batch of ``total_envs`` environments."""
    from .batched import BatchedMultiGrid
    rank, world = rank_world()
    start, count = shard_range(total_envs, rank, world)
    dev = int(os.environ.get("LOCAL_RANK", 0))
    return BatchedMultiGrid(spec, count, device=dev, env_id_offset=start, **kw)
