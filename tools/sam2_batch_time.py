import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from atlaspatch_amd.services.sam2_hip import Sam2HipPredictor
from atlaspatch_amd.services.segmentation import random_sam2_state_dict
pred = Sam2HipPredictor(random_sam2_state_dict(0), device="cuda")
rng = np.random.default_rng(0)
for B in (1, 2, 4, 8):
    imgs = torch.from_numpy(rng.integers(0, 256, (B, 1024, 1024, 3), dtype=np.uint8)).cuda()
    with torch.inference_mode():
        for _ in range(3): pred._graph_masks_device(imgs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): pred._graph_masks_device(imgs)
        e1.record(); torch.cuda.synchronize()
    print(f"B={B}: {e0.elapsed_time(e1)/10/B:.3f} ms per slide ({e0.elapsed_time(e1)/10:.2f} ms per forward)", flush=True)
