#!/usr/bin/env python3
"""ap_cv2_resize_u8 on the 40x -> 20x tile path: 2048 tiles read 512 x 512, resized to 256 x 256 (INTER_LINEAR = the exact
2x2 average OpenCV re-routes to), and a 1024 -> 256 INTER_LINEAR case; milliseconds and algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd.utils.resample import cv2_resize_device
dev = torch.device("cuda:0")
for n, s, o in ((2048, 512, 256), (512, 1024, 256), (2048, 300, 256)):
    t = torch.randint(0, 256, (n, s, s, 3), dtype=torch.uint8, device=dev)
    out = torch.empty((n, o, o, 3), dtype=torch.uint8, device=dev)
    for _ in range(3): cv2_resize_device(t, (o, o), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): cv2_resize_device(t, (o, o), out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = n * (s * s + o * o) * 3 / 1e9
    print(f"{n} x {s}^2 -> {o}^2: {ms:.3f} ms, {gb / ms * 1e3:.0f} GB/s algorithmic")
