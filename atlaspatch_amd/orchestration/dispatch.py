"""Device / rank dispatch on one MI355X node (BASELINE north star).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL on ROCm, ``gloo`` on CPU for
tests).  Slides are independent, so they are sharded one-per-rank with NO data-path collective;
the only exchange is the optional reassembly of the per-slide feature matrices on every rank:

    all_gather(row counts N_i)  ->  all_gather of [max N, D] padded blocks  ->  trim + concat

xGMI is point-to-point (7 links x ~153 GB/s per GPU): one padded all-gather moves each rank's
block over exactly one link per peer (a 152 881 x 768 f32 block = 470 MB ~ 3 ms/link), far off the
critical path next to seconds of embedding per slide, so a single large collective (not buckets)
is the right size.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch


def env_rank_world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (defaults 0, 1, 0)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(device: Optional[torch.device] = None, backend: Optional[str] = None) -> bool:
    """Initialise torch.distributed when launched with WORLD_SIZE > 1.  Returns True if it did."""
    import torch.distributed as dist
    rank, world, _ = env_rank_world()
    if world <= 1:
        return False
    if dist.is_initialized():
        return True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    kwargs = {"device_id": device} if backend == "nccl" and device is not None else {}
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return True


def shard(items: Sequence, rank: int, world_size: int) -> list:
    """Round-robin one-item-per-rank sharding (item i -> rank i % world_size)."""
    return list(items[rank::max(1, world_size)])


def gather_feature_matrix(local: torch.Tensor, group=None) -> list[torch.Tensor]:
    """All-gather-v of per-rank [N_i, D] float32 blocks; returns the list ordered by rank."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local]
    world = dist.get_world_size(group)
    device = local.device
    count = torch.tensor([local.shape[0], local.shape[1] if local.dim() == 2 else 0],
                         dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    rows = [int(c[0].item()) for c in counts]
    dim = max(int(c[1].item()) for c in counts)
    width = max(rows) if rows else 0
    padded = torch.zeros((width, dim), dtype=torch.float32, device=device)
    if local.shape[0]:
        padded[: local.shape[0]] = local.to(torch.float32)
    out = torch.empty((world * width, dim), dtype=torch.float32, device=device)
    if hasattr(dist, "all_gather_into_tensor") and device.type == "cuda":
        dist.all_gather_into_tensor(out, padded, group=group)
    else:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        out = torch.cat(parts, dim=0)
    return [out[r * width: r * width + rows[r]] for r in range(world)]
