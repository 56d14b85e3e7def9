#!/usr/bin/env python3
"""Race screen for the tile ring: N host tiles through the pinned 3-slot ring (slot reuse, copy-stream / compute-stream
hand-offs) must give exactly the features of direct forwards on the same tiles, run after run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "2")
import numpy as np, torch
from atlaspatch_amd.encoders import build_default_registry
from atlaspatch_amd.services.tile_ring import TileRing
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4500
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
ex = build_default_registry(device="cuda", dtype=torch.float16).create("vit_b_16")
dev = ex.device
host = np.random.default_rng(1).integers(0, 256, (N, 256, 256, 3), dtype=np.uint8)
ref = torch.empty((N, ex.embedding_dim), dtype=torch.float32, device=dev)
for lo in range(0, N, B):
    ex.forward_device(torch.from_numpy(host[lo:lo + B]).to(dev), ref[lo:lo + B])
ref = ref.cpu().numpy()
coords = np.stack([np.arange(N), np.zeros(N), np.full(N, 256), np.full(N, 256), np.zeros(N)], 1).astype(np.int32)
bad = 0
for workers in (4, 16):
    ring = TileRing(device=dev, batch=B, patch_size=256, slots=3, workers=workers)
    for r in range(runs):
        got = ring.run(coords, lambda x, y, rw, rh, lv: host[x], lambda t, o: ex.forward_device(t, o), ex.embedding_dim)
        if not np.array_equal(got, ref):
            rows = np.flatnonzero((got != ref).any(axis=1))
            bad += 1
            print(f"workers {workers} run {r}: {rows.size} rows differ (first {rows[:8].tolist()})", flush=True)
    ring.close()
print(f"ring: {bad} of {2 * runs} runs differ from the direct forwards")
sys.exit(1 if bad else 0)
