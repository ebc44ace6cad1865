"""Data parallelism over environments: one process per GPU, torch.distributed ("nccl" = RCCL on ROCm).

The path shards naturally: environments are independent, so each rank owns
E_total / world_size environments, their FSM state, their Philox substream (global env index =
env_offset + local index, so an env's trajectory does not depend on how the batch is sharded), its own
replay shard and its own sampled batch.  Networks are replicated.  The ONLY collective on the path is one
all-reduce (mean) of the flat trainable-gradient buffer per learner step (QMixLearner._allreduce_grads):
56 094 fp32 = 224 KB for 3j/4r, H=64 — latency-bound, so one flat buffer / one call, never per-tensor.
Every episode has the same number of filled steps, so mask.sum() is equal on all ranks and the mean of
the per-rank gradients equals the gradient of the reference loss over the global batch (qmix.py:194).
Rollout needs no communication at all.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch


def shard_range(total_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[lo, hi) global env indices owned by ``rank`` (contiguous, sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(total_envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(backend: str = None) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank); a no-op single-process world when WORLD_SIZE is unset."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def broadcast_parameters(modules, src: int = 0) -> None:
    """Make every rank start from rank ``src``'s weights (one flat broadcast per module)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    for m in modules:
        params = list(m.parameters())
        if not params:
            continue
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        dist.broadcast(flat, src=src)
        off = 0
        with torch.no_grad():   # (copy_ on the parameter itself, not on .data: its version counter moves, which is what
            for p in params:    # QMixLearner._body_is_shared watches)
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
