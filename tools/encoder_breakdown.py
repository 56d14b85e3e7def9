#!/usr/bin/env python3
"""usage: encoder_breakdown.py [encoder=uni_v1] [batch]   -- per-kernel-kind milliseconds of one forward (HIP events)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
import numpy as np, torch
from atlaspatch_amd.encoders import build_default_registry
name = sys.argv[1] if len(sys.argv) > 1 else "uni_v1"
ex = build_default_registry(device="cuda", dtype=torch.float16).create(name)
B = int(sys.argv[2]) if len(sys.argv) > 2 else ex.max_batch
dev = ex.device
tiles = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 256, 256, 3), dtype=np.uint8)).to(dev)
out = torch.empty((B, ex.embedding_dim), dtype=torch.float32, device=dev)
for _ in range(2):
    ex.forward_device(tiles, out)
torch.cuda.synchronize()
ex.vit.profile(True)
K = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    ex.forward_device(tiles, out)
e1.record(); torch.cuda.synchronize()
prof = ex.vit.profile_read()
ms = e0.elapsed_time(e1) / K
print(json.dumps({"encoder": name, "batch": B, "ms_per_forward": round(ms, 3), "tiles_per_s": round(B / ms * 1e3, 1),
                  "kinds_ms": {k: round(v[0] / K, 3) for k, v in prof.items()},
                  "launches": {k: v[1] // K for k, v in prof.items()}}))
