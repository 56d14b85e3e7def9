#!/usr/bin/env python3
"""Ten device border followings of one mask kind (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy import ndimage
from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
from atlaspatch_amd.utils.contours import DeviceContours
torch.zeros(1, device="cuda")
kind = sys.argv[1] if len(sys.argv) > 1 else "analytic"
rng = np.random.default_rng(0)
if kind == "analytic":
    m = analytic_mask(SynthSpec(width=100000, height=100000)).astype(np.float32)
else:
    f = ndimage.gaussian_filter(rng.standard_normal((1024, 1024)), 10.0)
    m = ((f > np.quantile(f, 0.55)) & (rng.random((1024, 1024)) < 0.97)).astype(np.float32)
for _ in range(10):
    DeviceContours(m, tissue_area_thresh=0.0005).close()
