#!/usr/bin/env python3
"""usage: forward_repeat.py [tiles=348] [iterations=600] [encoder=vit_b_16] [dtype=float16]
Race screen on the whole encoder: the same batch through the forward many times; every output must equal the first
bit for bit.  Prints the number of differing iterations and which rows differed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "2")
import numpy as np, torch
from atlaspatch_amd.encoders import build_default_registry
n = int(sys.argv[1]) if len(sys.argv) > 1 else 348
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 600
name = sys.argv[3] if len(sys.argv) > 3 else "vit_b_16"
dtype = getattr(torch, sys.argv[4]) if len(sys.argv) > 4 else torch.float16
ex = build_default_registry(device="cuda", dtype=dtype).create(name)
dev = ex.device
tiles = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).to(dev)
ref = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=dev)
ex.forward_device(tiles, ref)
torch.cuda.synchronize()
bad = 0
out = torch.empty_like(ref)
for it in range(iters):
    ex.forward_device(tiles, out)
    torch.cuda.synchronize()
    if not torch.equal(out, ref):
        rows = torch.nonzero((out != ref).any(dim=1)).flatten().tolist()
        bad += 1
        if bad <= 8:
            print(f"iteration {it}: rows {rows[:10]} differ, max |diff| {float((out - ref).abs().max()):.3e}", flush=True)
print(f"{name}: {bad} of {iters} forwards differ from the first")
sys.exit(1 if bad else 0)
