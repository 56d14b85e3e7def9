#!/usr/bin/env python3
"""Construction cost of the tile ring's pinned slots (3 x 2048 tiles of 256 x 256 x 3 = 1.2 GB of pinned host memory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd.services.tile_ring import TileRing
torch.cuda.init(); torch.zeros(1, device="cuda")
for it in range(2):
    t0 = time.perf_counter()
    r = TileRing(device=torch.device("cuda:0"), batch=2048, patch_size=256, slots=3, workers=8)
    t1 = time.perf_counter()
    r.close(); del r
    print(f"TileRing(batch 2048, 3 slots): {1e3 * (t1 - t0):.0f} ms")
t0 = time.perf_counter(); a = torch.empty((2048, 256, 256, 3), dtype=torch.uint8).pin_memory(); t1 = time.perf_counter()
b = torch.empty((2048, 256, 256, 3), dtype=torch.uint8, pin_memory=True); t2 = time.perf_counter()
print(f"one slot: empty().pin_memory() {1e3 * (t1 - t0):.0f} ms, empty(pin_memory=True) {1e3 * (t2 - t1):.0f} ms")
