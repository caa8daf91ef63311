"""Data-parallel glue: one process per GPU, gradients averaged with ONE collective per optimizer.

The reference has no multi-GPU code.  The path shards cleanly by image: InstanceNorm, window
attention and the discriminator are all per-sample and the losses are batch means, so averaging the per-rank
gradients of equal shards reproduces the global-batch gradient.  ``FlatAdam`` keeps all gradients of
an optimizer in one contiguous buffer, so the exchange is a single all-reduce (RCCL over xGMI with the ``nccl``
backend; ``gloo`` on CPU in the unit tests) -- latency-bound at ~1-3 MB, hence one big message, not buckets.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend: str | None = None) -> int:
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun) if WORLD_SIZE > 1; returns local rank."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)
    return local


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place average of a flat gradient buffer over all ranks (no-op for a single process)."""
    ws = world_size()
    if ws == 1:
        return flat
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / ws)
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    if world_size() > 1:
        dist.broadcast(flat, src=src)
        from . import ops
        ops.bump_pack_epoch()  # the collective wrote parameter memory without touching torch's version counters
    return flat


def shard(batch: torch.Tensor) -> torch.Tensor:
    """This rank's equal slice of a global batch (dim 0 must divide by the world size)."""
    ws, r = world_size(), rank()
    if ws == 1:
        return batch
    if batch.shape[0] % ws:
        raise ValueError(f"global batch {batch.shape[0]} does not divide over {ws} ranks")
    per = batch.shape[0] // ws
    return batch[r * per:(r + 1) * per]
