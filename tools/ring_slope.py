import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
from atlaspatch_amd.encoders import build_default_registry
from atlaspatch_amd.services.tile_ring import TileRing
dev = torch.device("cuda:0")
ex = build_default_registry(device=dev, dtype=torch.float16).create("vit_b_16")
B = 2048
rng = np.random.default_rng(0)
host = rng.integers(0, 256, (4096, 256, 256, 3), dtype=np.uint8)
res = {}
for mult in (8, 16, 32):
    N = mult * B
    coords = np.stack([np.arange(N) % 4096, np.zeros(N), np.full(N, 256), np.full(N, 256), np.zeros(N)], 1).astype(np.int32)
    ring = TileRing(device=dev, batch=B, patch_size=256, slots=3, workers=32)
    fwd = lambda t, o: ex.vit.forward_u8(t, ex.mean, ex.std, o)
    ring.run(coords[:2 * B], lambda x, y, rw, rh, lv: host[x], fwd, 768)
    t0 = time.perf_counter()
    ring.run(coords, lambda x, y, rw, rh, lv: host[x], fwd, 768)
    dt = time.perf_counter() - t0
    ring.close()
    res[mult] = dt
    print(f"{N} tiles: {dt:.3f} s = {N/dt:.0f} tiles/s", flush=True)
print("slope 16->32 batches:", 16 * B / (res[32] - res[16]), "tiles/s steady state")
