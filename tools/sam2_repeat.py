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
from PIL import Image
img1024 = np.asarray(Image.fromarray(img).resize((1024, 1024), Image.Resampling.BILINEAR))
ref_logits = pred.predict_logits(img1024).clone()            # the float32 logits, bit for bit (a thresholded mask hides small races)
bad = bad_logits = 0
for it in range(iters):
    got = pred.predict_image(img)
    if not np.array_equal(np.asarray(got), np.asarray(ref)):
        bad += 1
    if not torch.equal(pred.predict_logits(img1024), ref_logits):
        bad_logits += 1
print(f"sam2: {bad} of {iters} predictions and {bad_logits} of {iters} logit maps differ from the first")
sys.exit(1 if bad or bad_logits else 0)
