#!/usr/bin/env python3
"""Where embed_matrix's time goes on the device-tile-source path (100 000^2 synthetic slide): per-batch tile synthesis,
forward, the final D2H, first-call costs (workspace allocation).  Synchronises after every piece: a probe, not a rate."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
import numpy as np, torch
from atlaspatch_amd.core.wsi.synth_pixels import SynthSpec, analytic_mask
from atlaspatch_amd.core.wsi.synth_wsi import SynthWSI
from atlaspatch_amd.encoders import build_default_registry
from atlaspatch_amd.services.extraction import coords_from_mask
side = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
def T(): torch.cuda.synchronize(); return time.perf_counter()
t0 = T()
ex = build_default_registry(device=dev, dtype=torch.float16).create("vit_b_16")
t1 = T(); print("encoder_create", round(t1 - t0, 4))
spec = SynthSpec(width=side, height=side, seed=1234)
coords, _ = coords_from_mask(analytic_mask(spec), level0_wh=(side, side), downsamples=list(spec.downsamples), src_mag=20, tgt_mag=20,
                             patch_size=256, step_size=None, tissue_thresh=0.0)
with tempfile.TemporaryDirectory() as tmp:
    p = os.path.join(tmp, "s.synth")
    json.dump({"width": side, "height": side, "seed": 1234, "mag": 20, "mpp": 0.5, "downsamples": [1, 4, 16]}, open(p, "w"))
    wsi = SynthWSI(p)
    n = coords.shape[0]
    for rep in range(2):
        t0 = T()
        out = torch.empty((n, 768), dtype=torch.float32, device=dev)
        ta = T()
        src = fwd = 0.0
        for lo in range(0, n, 2048):
            a = T()
            tiles = wsi.extract_batch_device(coords[lo:lo + 2048], dev, 256)
            b = T()
            ex.forward_device(tiles, out[lo:lo + tiles.shape[0]])
            c = T()
            src += b - a; fwd += c - b
            if lo == 0: print("  first batch: source", round(b - a, 4), "forward", round(c - b, 4))
        td = T()
        host = out.cpu().numpy()
        te = T()
        print(f"rep {rep}: alloc_out {ta - t0:.4f} source {src:.4f} forward {fwd:.4f} d2h {te - td:.4f} total {te - t0:.4f} -> {n / (te - t0):.0f} tiles/s; "
              f"forward-only {n / fwd:.0f}")
    # unsynchronised loop as the product runs it
    t0 = T()
    out = torch.empty((n, 768), dtype=torch.float32, device=dev)
    for lo in range(0, n, 2048):
        tiles = wsi.extract_batch_device(coords[lo:lo + 2048], dev, 256)
        ex.forward_device(tiles, out[lo:lo + tiles.shape[0]])
    tq = time.perf_counter()
    host = out.cpu().numpy()
    t1 = T()
    print(f"async loop: enqueue {tq - t0:.4f} total {t1 - t0:.4f} -> {n / (t1 - t0):.0f} tiles/s")
