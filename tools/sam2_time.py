#!/usr/bin/env python3
"""Latency of one SAM2 Hiera-T tissue segmentation (1024 x 1024 thumbnail) on the HIP operator set:
launch by launch (image encoder / mask decoder) and as the captured hipGraph (what predict_image replays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
from atlaspatch_amd.services.segmentation import random_sam2_state_dict
pred = Sam2HipPredictor(random_sam2_state_dict(0), device="cuda:0")
img = np.random.default_rng(0).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
d = torch.from_numpy(img).cuda()
for _ in range(2):
    pred.mask_logits(*pred.image_features(d))
torch.cuda.synchronize()
t0 = time.perf_counter()
f = pred.image_features(d); torch.cuda.synchronize(); t1 = time.perf_counter()
pred.mask_logits(*f); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"launch by launch: image encoder {1e3*(t1-t0):.2f} ms, mask decoder {1e3*(t2-t1):.2f} ms")
pred.predict_image(img)                       # captures the graph
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    pred._graph.replay()
ev1.record(); torch.cuda.synchronize()
print(f"hipGraph replay: {ev0.elapsed_time(ev1)/10:.2f} ms per slide (device time)")
t0 = time.perf_counter()
for _ in range(5):
    pred.predict_image(img)
print(f"predict_image (host to host, graph): {1e3*(time.perf_counter()-t0)/5:.2f} ms")
