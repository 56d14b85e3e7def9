#!/usr/bin/env python3
"""A/B of the ring's decode-thread pinning (ATLASPATCH_PIN_THREADS) in one process on one box: PCIe-inclusive ring rate from an
in-memory source and from the native renderer (read_chunk hook), pinned vs unpinned, interleaved."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
from atlaspatch_amd.encoders import build_default_registry
from atlaspatch_amd.services.tile_ring import TileRing
dev = torch.device("cuda:0")
ex = build_default_registry(device=dev, dtype=torch.float16).create("vit_b_16")
B, N = 2048, 8192
host = np.random.default_rng(0).integers(0, 256, (N, 256, 256, 3), dtype=np.uint8)
coords = np.stack([np.arange(N), np.zeros(N), np.full(N, 256), np.full(N, 256), np.zeros(N)], 1).astype(np.int32)
out = {}
for rep in range(2):
    for pin in ("1", "0"):
        for workers in (16, 32, 64):
            os.environ["ATLASPATCH_PIN_THREADS"] = pin
            ring = TileRing(device=dev, batch=B, patch_size=256, slots=3, workers=workers)
            fwd = lambda t, o: ex.forward_device(t, o)
            read = lambda x, y, rw, rh, lv: host[x]
            ring.run(coords[:B], read, fwd, ex.embedding_dim)
            t0 = time.perf_counter(); ring.run(coords, read, fwd, ex.embedding_dim); dt = time.perf_counter() - t0
            ring.close()
            out.setdefault(f"pin{pin}_w{workers}", []).append(round(N / dt))
print(json.dumps(out))
