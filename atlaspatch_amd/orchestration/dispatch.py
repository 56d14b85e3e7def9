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


GATHER_ALGORITHMS = ("allgather", "pairs")


def gather_algorithm() -> str:
    """``ATLASPATCH_GATHER_ALGO``: ``allgather`` (default: one padded ``all_gather_into_tensor``, whatever schedule RCCL
    picks) or ``pairs`` (every rank sends its exact block to every peer, one point-to-point transfer per xGMI link, no
    padding).  Same result; the first 8-GPU run can A/B them against the ~3 ms / link estimate with one variable."""
    algo = os.environ.get("ATLASPATCH_GATHER_ALGO", "") or "allgather"
    if algo not in GATHER_ALGORITHMS:
        raise ValueError(f"ATLASPATCH_GATHER_ALGO={algo!r}: expected one of {GATHER_ALGORITHMS}")
    return algo


def gather_feature_matrix(local: torch.Tensor, group=None, algorithm: Optional[str] = None) -> list[torch.Tensor]:
    """All-gather-v of per-rank [N_i, D] float32 blocks; returns the list ordered by rank.  A rank without rows may pass
    any width (``[0, 0]`` included): the width is the largest any rank reports.  ``algorithm``: see ``gather_algorithm``."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local]
    algorithm = algorithm or gather_algorithm()
    if algorithm not in GATHER_ALGORITHMS:
        raise ValueError(f"gather_feature_matrix: algorithm {algorithm!r}, expected one of {GATHER_ALGORITHMS}")
    world, me = dist.get_world_size(group), dist.get_rank(group)
    device = local.device
    count = torch.tensor([local.shape[0], local.shape[1] if local.dim() == 2 and local.shape[0] else 0],
                         dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    rows = [int(c[0].item()) for c in counts]
    dims = {int(c[1].item()) for c in counts if int(c[0].item())}
    if len(dims) > 1:
        raise ValueError(f"gather_feature_matrix: ranks hold blocks of different widths {sorted(dims)}")
    dim = dims.pop() if dims else (local.shape[1] if local.dim() == 2 else 0)
    mine = local.to(torch.float32).contiguous() if local.shape[0] else torch.zeros((0, dim), dtype=torch.float32, device=device)
    if algorithm == "pairs":
        # all-pairs exchange: rank r's exact [N_r, D] block goes to every peer as one point-to-point transfer; a rank without
        # rows neither sends nor is waited for.  On xGMI every (sender, receiver) pair has its own link.
        parts = [mine if r == me else torch.empty((rows[r], dim), dtype=torch.float32, device=device) for r in range(world)]
        ops = []
        for off in range(1, world):                  # peer order rotated per rank: no two ranks start on the same target
            peer = (me + off) % world
            src = (me - off) % world
            if rows[me]:
                ops.append(dist.P2POp(dist.isend, mine, peer if group is None else dist.get_global_rank(group, peer), group))
            if rows[src]:
                ops.append(dist.P2POp(dist.irecv, parts[src], src if group is None else dist.get_global_rank(group, src), group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return parts
    width = max(rows) if rows else 0
    padded = torch.zeros((width, dim), dtype=torch.float32, device=device)
    if rows[me]:
        padded[: rows[me]] = mine
    out = torch.empty((world * width, dim), dtype=torch.float32, device=device)
    if hasattr(dist, "all_gather_into_tensor") and device.type == "cuda":
        dist.all_gather_into_tensor(out, padded, group=group)
    else:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        out = torch.cat(parts, dim=0)
    return [out[r * width: r * width + rows[r]] for r in range(world)]


def _local_feature_block(path, name: str, cache: Optional[dict]):
    """float32 [N, D] of ``features/<name>`` for one slide, or None (with the reason) when this rank cannot contribute it:
    file or dataset missing (feature phase skipped because another process held the lock, an extractor that failed) or a
    row count that disagrees with ``coords``.  Never raises: a rank that raised here would leave its peers blocked in the
    collectives that follow."""
    import numpy as np
    from ..utils.h5 import h5
    key = (str(path), name)
    if cache is not None and key in cache:
        return cache[key], None
    try:
        with h5.File(str(path), "r") as fh:
            if "features" not in fh or name not in fh["features"]:
                return None, "no such feature set"
            feats = np.asarray(fh["features"][name][:], dtype=np.float32)
            if feats.ndim != 2 or feats.shape[0] != fh["coords"].shape[0]:
                return None, f"shape {feats.shape} does not match {fh['coords'].shape[0]} coords rows"
        return feats, None
    except Exception as exc:  # noqa: BLE001
        return None, f"unreadable: {exc}"


def gather_run_features(h5_paths: Sequence, extractor_names: Sequence[str], out_root, *, device=None,
                        cache: Optional[dict] = None) -> dict:
    """Reassemble the feature matrices of a rank-sharded run on every rank (the one exchange step of the north star).

    ``h5_paths``: the ``<stem>.h5`` outputs of ALL slides assigned to this rank, in processing order -- including slides a
    ``--skip-existing`` rerun found complete on disk.  ``cache``: ``{(h5 path, extractor): float32 [N, D] array or device
    tensor}`` of feature blocks this process computed in this run (``PatchFeatureEmbeddingService.feature_blocks``); blocks
    found there are not read back from disk.  Per extractor: the rank's ``[sum N_i, D]`` block (its slides concatenated)
    goes through ONE all-gather-v (``gather_feature_matrix``); rank 0 writes ``<out_root>/features_all/<extractor>.npy``
    ([total, D] float32, ranks in order, slides in each rank's order) and ``<extractor>.index.json`` (slide stem, rank,
    first row, rows; slides that could not contribute are listed with ``"rows": 0`` and the reason).  A slide whose feature
    set is missing is skipped with a warning on its own rank -- nothing raises before or between the collectives, so no
    rank can leave the others waiting.  Returns ``{extractor: [total, D] tensor}`` on every rank.  A reference run has no
    such file: this is the MI355X multi-GPU addition, enabled by ``ATLASPATCH_GATHER_FEATURES=1`` under
    ``torch.distributed.run``."""
    import json
    import logging
    from pathlib import Path
    import numpy as np
    import torch.distributed as dist
    log = logging.getLogger("atlaspatch_amd.dispatch")
    rank, world, _ = env_rank_world()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    merged: dict = {}
    for name in extractor_names:
        blocks, index = [], []
        for path in h5_paths:
            feats, why = _local_feature_block(path, name, cache)
            entry = {"slide": Path(str(path)).stem, "rank": rank, "rows": 0}
            if feats is None:
                log.warning("gather: %s has no usable '%s' features (%s); skipped", path, name, why)
                entry["skipped"] = why
            else:
                block = feats if torch.is_tensor(feats) else torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32))
                entry["rows"] = int(block.shape[0])
                blocks.append(block.to(dev, dtype=torch.float32))
            index.append(entry)
        dim = blocks[0].shape[1] if blocks else 0
        local = torch.cat(blocks, 0) if blocks else torch.zeros((0, dim), dtype=torch.float32, device=dev)
        parts = gather_feature_matrix(local)
        width = max((p.shape[1] for p in parts if p.dim() == 2 and p.shape[0]), default=dim)
        whole = torch.cat([p if p.shape[0] else p.reshape(0, width) for p in parts], 0) if parts else local
        merged[name] = whole
        all_index = [index]
        if dist.is_available() and dist.is_initialized() and world > 1:
            all_index = [None] * world
            dist.all_gather_object(all_index, index)
        if rank == 0:
            out_dir = Path(out_root) / "features_all"
            out_dir.mkdir(parents=True, exist_ok=True)
            np.save(out_dir / f"{name}.npy", whole.cpu().numpy())
            rows, first = [], 0
            for per_rank in all_index:
                for item in per_rank:
                    rows.append({**item, "first_row": first})
                    first += item["rows"]
            (out_dir / f"{name}.index.json").write_text(json.dumps(rows, indent=1))
    return merged
