#!/usr/bin/env python3
"""Race screen for the SAM2 image path: the same image through the HIP predictor many times, masks bit-equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
from atlaspatch_amd.services.segmentation import random_sam2_state_dict
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pred = Sam2HipPredictor(random_sam2_state_dict(3), device=torch.device("cuda:0"))
rng = np.random.default_rng(1)
img = rng.integers(0, 256, (700, 900, 3), dtype=np.uint8)
img[100:500, 200:700] = (img[100:500, 200:700] // 3 + 120).astype(np.uint8)
ref = pred.predict_image(img)
bad = 0
for it in range(iters):
    got = pred.predict_image(img)
    if not np.array_equal(np.asarray(got), np.asarray(ref)):
        bad += 1
print(f"sam2: {bad} of {iters} predictions differ from the first")
sys.exit(1 if bad else 0)
