import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd.encoders.vit import build_hip_vit_extractor
from atlaspatch_amd.services.tile_ring import TileRing
dev = torch.device("cuda:0")
ex = build_hip_vit_extractor(name="ring", arch="vit_b_16", depth=1, device=dev, dtype=torch.float16, random_init_seed=3)
rng = np.random.default_rng(0)
n = 70
tiles = rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
coords = np.stack([np.arange(n), np.zeros(n), np.full(n, 256), np.full(n, 256), np.zeros(n)], 1).astype(np.int32)
want = ex.extract_batch(list(tiles), batch_size=32)
for trial in range(3):
    ring = TileRing(device=dev, batch=16, patch_size=256, slots=2, workers=3)
    got = ring.run(coords, lambda x, y, rw, rh, lv: tiles[x], lambda t, o: ex.vit.forward_u8(t, ex.mean, ex.std, o), 768)
    ring.close()
    d = np.abs(got - want).max(1)
    print("trial", trial, "equal:", np.array_equal(got, want), "rows differing:", np.where(d > 0)[0], "max", d.max(), "nan rows", np.where(np.isnan(got).any(1))[0])
