#!/usr/bin/env python3
"""Features must not depend on how a slide's tiles are cut into device batches (bit for bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "2")
import numpy as np, torch
from atlaspatch_amd.encoders import build_default_registry
bad = 0
for name, n in (("vit_b_16", 700), ("uni_v1", 300), ("conch_v1", 40)):
    ex = build_default_registry(device="cuda", dtype=torch.float16).create(name)
    dev = ex.device
    tiles = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).to(dev)
    ref = torch.empty((n, ex.embedding_dim), dtype=torch.float32, device=dev)
    ex.forward_device(tiles, ref)
    for chunk in (1, 7, 64, 255, 256, 257, 300):
        out = torch.empty_like(ref)
        for lo in range(0, n, chunk):
            ex.forward_device(tiles[lo:lo + chunk], out[lo:lo + chunk])
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        bad += not same
        print(f"{name}: chunks of {chunk:3d} vs one batch of {n}: {'bit-identical' if same else 'DIFFERS, max ' + str(float((out - ref).abs().max()))}", flush=True)
    ex.cleanup()
sys.exit(1 if bad else 0)
