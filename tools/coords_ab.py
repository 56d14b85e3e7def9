#!/usr/bin/env python3
"""Coordinate generation (contours -> grid scan -> rows) for synthetic slides and for a SAM2-like ragged mask: the
row-bucketed grid kernel vs the full scan (AP_GRID_FLAGS_LEGACY=1), rows equal, warm median of 10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
from atlaspatch_amd.services.extraction import coords_from_mask
torch.zeros(1, device="cuda")
def ragged(seed, n=1024):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    f = ndimage.gaussian_filter(rng.standard_normal((n, n)), 3.0)
    return ((f > np.quantile(f, 0.45)) & (rng.random((n, n)) < 0.97)).astype(np.float32)
cases = [("analytic 40k", analytic_mask(SynthSpec(width=40000, height=40000)), 40000),
         ("analytic 100k", analytic_mask(SynthSpec(width=100000, height=100000)), 100000),
         ("ragged 1024^2 mask, 100k slide", ragged(0), 100000)]
for name, mask, side in cases:
    kw = dict(level0_wh=(side, side), downsamples=[1.0, 4.0, 16.0], src_mag=20, tgt_mag=20, patch_size=256, step_size=None, tissue_thresh=0.0)
    out = {}
    for mode in ("bucketed", "legacy"):
        if mode == "legacy": os.environ["AP_GRID_FLAGS_LEGACY"] = "1"
        else: os.environ.pop("AP_GRID_FLAGS_LEGACY", None)
        rows, _ = coords_from_mask(mask, **kw)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); coords_from_mask(mask, **kw); ts.append(time.perf_counter() - t0)
        out[mode] = (rows, sorted(ts)[5])
    os.environ.pop("AP_GRID_FLAGS_LEGACY", None)
    print(f"{name}: rows {out['bucketed'][0].shape[0]}  equal {np.array_equal(out['bucketed'][0], out['legacy'][0])}  "
          f"bucketed {out['bucketed'][1]*1e3:.2f} ms  legacy {out['legacy'][1]*1e3:.2f} ms", flush=True)
