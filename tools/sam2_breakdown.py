#!/usr/bin/env python3
"""Where one SAM2 Hiera-T forward spends its device time, launch by launch: every ap_sgemm is timed with HIP events and
grouped by shape (batch x M x N x K, NT / NN, activation); the other operators by name."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd import _lib
from atlaspatch_amd.services import sam2_hip
from atlaspatch_amd.services.segmentation import random_sam2_state_dict
pred = sam2_hip.Sam2HipPredictor(random_sam2_state_dict(0), device="cuda:0")
img = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)).cuda()
for _ in range(2): pred.mask_logits(*pred.image_features(img))
torch.cuda.synchronize()
events = []
lib = pred.lib
class Timed:
    def __init__(self, inner): self.inner = inner
    def __getattr__(self, name):
        fn = getattr(self.inner, name)
        if not name.startswith("ap_") or name in ("ap_last_error",): return fn
        def wrapped(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a); e1.record()
            if name == "ap_sgemm":
                key = f"sgemm b{a[7]} M{a[8]} N{a[9]} K{a[10]} {'NN' if a[6] else 'NT'} act{a[13]}{' +res' if a[14] else ''}"
                flop = 2.0 * a[7] * a[8] * a[9] * a[10]
            elif name == "ap_sattention_f32":
                nb, heads, tq, tk, d = a[6], a[7], a[8], a[9], a[10]
                key = f"sattention nb{nb} heads{heads} tq{tq} tk{tk} d{d}"
                flop = 4.0 * nb * heads * tq * tk * d
            elif name == "ap_gemm":
                key = f"gemm128(f32) M{a[6]} N{a[7]} K{a[8]} epi{a[1]}"
                flop = 2.0 * a[6] * a[7] * a[8]
            else:
                key, flop = name, 0.0
            events.append((key, flop, e0, e1)); return r
        return wrapped
pred.lib = Timed(lib)
pred.mask_logits(*pred.image_features(img))
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, flop, e0, e1 in events:
    ms = e0.elapsed_time(e1)
    c = agg.setdefault(key, [0, 0.0, 0.0]); c[0] += 1; c[1] += ms; c[2] += flop
tot = sum(v[1] for v in agg.values())
print(f"total {tot:.3f} ms over {len(events)} launches (event-timed, includes launch gaps)")
for key, (n, ms, flop) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    tf = flop / ms / 1e9 if flop else 0
    print(f"{ms:7.3f} ms  {n:4d}x  {tf:6.1f} TF/s  {key}")
