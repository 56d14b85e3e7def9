#!/usr/bin/env python3
"""Border following on the device (contours_device.hip) vs on the host (contours.cpp): ms per mask for the three mask kinds the
pipeline meets, and a repeat screen (the device form uses a lock-free union-find: equal results over many repeats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy import ndimage
from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
from atlaspatch_amd.utils.contours import DeviceContours
torch.zeros(1, device="cuda")
rng = np.random.default_rng(0)
f = ndimage.gaussian_filter(rng.standard_normal((1024, 1024)), 10.0)
masks = {"analytic_100k": analytic_mask(SynthSpec(width=100000, height=100000)).astype(np.float32),
         "blobs_ragged": ((f > np.quantile(f, 0.55)) & (rng.random((1024, 1024)) < 0.97)).astype(np.float32),
         "noise": (rng.random((1024, 1024)) < 0.5).astype(np.float32)}
def run(mask, host):
    if host: os.environ["AP_CONTOURS_HOST"] = "1"
    else: os.environ.pop("AP_CONTOURS_HOST", None)
    dc = DeviceContours(mask, tissue_area_thresh=0.0005)
    t, h = dc.as_lists()
    dc.close()
    return t, h
def same(a, b):
    return len(a[0]) == len(b[0]) and all(np.array_equal(x, y) for x, y in zip(a[0], b[0])) and \
        all(len(p) == len(q) and all(np.array_equal(x, y) for x, y in zip(p, q)) for p, q in zip(a[1], b[1]))
for name, m in masks.items():
    ref = run(m, True)
    out = {}
    for host in (True, False):
        run(m, host)
        t0 = time.perf_counter()
        for _ in range(10):
            os.environ["AP_CONTOURS_HOST"] = "1" if host else ""
            if not host: os.environ.pop("AP_CONTOURS_HOST")
            dc = DeviceContours(m, tissue_area_thresh=0.0005); dc.close()
        out["host" if host else "device"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
    bad = sum(0 if same(run(m, False), ref) else 1 for _ in range(40))
    print(name, "tissue", len(ref[0]), "holes", sum(len(x) for x in ref[1]), "points", sum(len(x) for x in ref[0]), "ms", out, "repeats differing", bad, "of 40", flush=True)
