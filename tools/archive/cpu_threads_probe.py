#!/usr/bin/env python3
"""How fast is the CPU oracle (ViT-B/16 fp32, batch 32) at different torch thread counts on this host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print("cpus", os.cpu_count(), "default torch threads", torch.get_num_threads())
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    torch.set_num_threads(th)
    t0 = time.perf_counter()
    out, hf, patches, rate = bench.cpu_baseline(64, seed=0)
    print(f"threads {th:4d}: {rate:7.2f} patches/s  (call incl. weights {time.perf_counter() - t0:.1f} s)", flush=True)
