#!/usr/bin/env python3
"""Device border following vs the host form on one golden mask: where do they differ?"""
import os, sys, json, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers
from atlaspatch_amd.utils.contours import mask_to_contours
name = sys.argv[1] if len(sys.argv) > 1 else "noise_field"
case = helpers.load_coords_cases()[name]
mask, thr = case["mask"], case["info"]["config"]["tissue_thresh"]
if os.environ.get("AP_CONTOURS_HOST"):
    t, h = mask_to_contours(mask, tissue_area_thresh=thr)
    np.save(sys.argv[2], np.array([json.dumps([[a.reshape(-1, 2).tolist() for a in t], [[b.reshape(-1, 2).tolist() for b in hs] for hs in h]])]))
    sys.exit(0)
t, h = mask_to_contours(mask, tissue_area_thresh=thr)
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "h.npy")
    subprocess.run([sys.executable, __file__, name, out], check=True, env=dict(os.environ, AP_CONTOURS_HOST="1"))
    wt, wh = json.loads(str(np.load(out)[0]))
print("tissue", len(t), len(wt), "holes", sum(len(x) for x in h), sum(len(x) for x in wh))
for i, (a, b) in enumerate(zip(t, wt)):
    if not np.array_equal(a.reshape(-1, 2), np.asarray(b)): print("tissue differs", i, len(a), len(b))
for i, (hs, ws) in enumerate(zip(h, wh)):
    if len(hs) != len(ws) or any(not np.array_equal(x.reshape(-1, 2), np.asarray(y)) for x, y in zip(hs, ws)):
        print("holes differ at tissue", i, [len(x) for x in hs], [len(y) for y in ws], [x.reshape(-1, 2)[0].tolist() for x in hs], [y[0] for y in ws])
