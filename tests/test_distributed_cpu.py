"""world_size-2 gloo tests of the rank dispatch (slide sharding + feature all-gather-v)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    import torch.distributed as dist
    from atlaspatch_amd.orchestration.dispatch import gather_feature_matrix, shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        slides = [f"s{i}" for i in range(5)]
        mine = shard(slides, rank, world)
        assert mine == slides[rank::world]
        rows = [3, 7][rank]                                   # ragged: different N per rank
        local = torch.full((rows, 4), float(rank + 1)) + torch.arange(rows).view(-1, 1)
        parts = gather_feature_matrix(local)
        assert [p.shape for p in parts] == [(3, 4), (7, 4)]
        for r, p in enumerate(parts):
            want = torch.full(([3, 7][r], 4), float(r + 1)) + torch.arange([3, 7][r]).view(-1, 1)
            assert torch.equal(p, want)
        empty = gather_feature_matrix(torch.zeros((0, 4)) if rank == 0 else local)   # one empty shard
        assert empty[0].shape == (0, 4) and empty[1].shape == (7, 4)
        np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0.npy").exists() and (tmp_path / "ok1.npy").exists()


def _worker_world4(rank, world, port, tmp):
    import torch.distributed as dist
    from atlaspatch_amd.orchestration.dispatch import GATHER_ALGORITHMS, gather_feature_matrix
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows_of = [0, 1, 3, 58938]                            # an empty rank, two tiny ones, one 100 000^2 slide's rows
        D = 768

        def block(r):
            g = torch.Generator().manual_seed(100 + r)
            return torch.randn((rows_of[r], D), generator=g)
        # the empty rank reports a DIFFERENT width (a rank whose slides all failed knows no D): [0, 0], and [0, 5]
        for empty_shape in ((0, 0), (0, 5)):
            local = block(rank) if rows_of[rank] else torch.zeros(empty_shape)
            got = {}
            for algo in GATHER_ALGORITHMS:
                parts = gather_feature_matrix(local, algorithm=algo)
                assert [tuple(p.shape) for p in parts] == [(n, D) for n in rows_of], (algo, [p.shape for p in parts])
                for r, p in enumerate(parts):
                    assert p.dtype == torch.float32 and torch.equal(p, block(r)), (algo, r)
                got[algo] = parts
        # every rank empty: nothing to exchange, widths stay what the caller passed
        for algo in GATHER_ALGORITHMS:
            parts = gather_feature_matrix(torch.zeros((0, 16)), algorithm=algo)
            assert [tuple(p.shape) for p in parts] == [(0, 16)] * world
        # ranks that disagree on the width of NON-empty blocks are an error on every rank, before any payload moves
        with pytest.raises(ValueError, match="different widths"):
            gather_feature_matrix(torch.zeros((2, 4 + rank % 2)))
        np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_gather_world4_real_row_counts_both_algorithms(tmp_path, monkeypatch):
    """Four ranks holding 0 / 1 / 3 / 58 938 rows of 768 floats (an empty rank that reports another width included): the padded
    all-gather and the all-pairs point-to-point exchange (ATLASPATCH_GATHER_ALGO) both return every rank's exact block."""
    from atlaspatch_amd.orchestration import dispatch
    port = _free_port()
    mp.spawn(_worker_world4, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert all((tmp_path / f"ok{r}.npy").exists() for r in range(4))
    monkeypatch.setenv("ATLASPATCH_GATHER_ALGO", "pairs")
    assert dispatch.gather_algorithm() == "pairs"
    monkeypatch.setenv("ATLASPATCH_GATHER_ALGO", "ring-of-fire")
    with pytest.raises(ValueError):
        dispatch.gather_algorithm()
    monkeypatch.delenv("ATLASPATCH_GATHER_ALGO")
    assert dispatch.gather_algorithm() == "allgather"


def test_runner_shards_slides_one_per_rank(tmp_path):
    import json
    from pathlib import Path
    from atlaspatch_amd.core.config import (AppConfig, ExtractionConfig, OutputConfig, ProcessingConfig,
                                            SegmentationConfig)
    from atlaspatch_amd.orchestration.runner import ProcessingRunner
    for i in range(5):
        json.dump({"width": 4096, "height": 4096}, open(tmp_path / f"slide{i}.synth", "w"))
    yaml = tmp_path / "seg.yaml"
    yaml.write_text("model: {}\n")
    seen = []
    for rank in range(2):
        cfg = AppConfig(processing=ProcessingConfig(input_path=tmp_path),
                        segmentation=SegmentationConfig(checkpoint_path=None, config_path=yaml, device="cpu"),
                        extraction=ExtractionConfig(patch_size=256, target_magnification=20),
                        output=OutputConfig(output_root=tmp_path / "out"))
        runner = ProcessingRunner(cfg, None, None, None, None, None, rank=rank, world_size=2)
        seen.append([s.path.name for s in runner.discover_slides()])
    assert seen[0] == ["slide0.synth", "slide2.synth", "slide4.synth"]
    assert seen[1] == ["slide1.synth", "slide3.synth"]


def _gather_worker(rank, world, port, tmp):
    import json
    import torch.distributed as dist
    from atlaspatch_amd.orchestration.dispatch import gather_run_features
    from atlaspatch_amd.utils.h5 import h5
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 owns slides a (3 rows) and c (2 rows), rank 1 owns b (4 rows) and d; features = slide id + row / 10.
        # d has coords but NO feature set (its feature phase was skipped: lock held elsewhere): it must be skipped on rank 1
        # without raising -- a raise before the collectives would leave rank 0 waiting forever.  c's block comes from the
        # in-run cache (its H5 holds no features at all), like a block the embedding service just computed.
        mine = {0: [("a", 3, 1.0), ("c", 2, 3.0)], 1: [("b", 4, 2.0), ("d", 6, None)]}[rank]
        paths, cache = [], {}
        for stem, rows, base in mine:
            p = os.path.join(tmp, f"{stem}.h5")
            feats = None if base is None else (base + np.arange(rows)[:, None] / 10 + np.zeros((rows, 5))).astype(np.float32)
            with h5.File(p, "w") as fh:
                fh.create_dataset("coords", data=np.zeros((rows, 5), np.int32))
                if feats is not None and stem != "c":
                    fh.create_group("features").create_dataset("enc", data=feats)
            if stem == "c":
                cache[(p, "enc")] = feats
            paths.append(p)
        merged = gather_run_features(paths, ["enc"], os.path.join(tmp, "out"), cache=cache)
        assert merged["enc"].shape == (9, 5)
        assert torch.allclose(merged["enc"][:, 0], torch.tensor([1.0, 1.1, 1.2, 3.0, 3.1, 2.0, 2.1, 2.2, 2.3]))
        dist.barrier()
        if rank == 0:
            whole = np.load(os.path.join(tmp, "out", "features_all", "enc.npy"))
            index = json.load(open(os.path.join(tmp, "out", "features_all", "enc.index.json")))
            assert whole.shape == (9, 5) and whole.dtype == np.float32
            assert [(r["slide"], r["rank"], r["first_row"], r["rows"]) for r in index] == \
                [("a", 0, 0, 3), ("c", 0, 3, 2), ("b", 1, 5, 4), ("d", 1, 9, 0)]
            assert "skipped" in index[3] and "skipped" not in index[0]
        np.save(os.path.join(tmp, f"gok{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_gather_run_features_world2(tmp_path):
    """The run-level reassembly (ATLASPATCH_GATHER_FEATURES): per-rank H5 feature sets -> one [total, D] matrix + index."""
    port = _free_port()
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "gok0.npy").exists() and (tmp_path / "gok1.npy").exists()
