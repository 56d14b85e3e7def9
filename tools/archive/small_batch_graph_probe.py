#!/usr/bin/env python3
"""extract_batch's call shape (32 host patches): where the 3.3 ms go, and what a captured hipGraph of the forward returns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
from atlaspatch_amd.encoders import build_default_registry
dev = torch.device("cuda:0")
ex = build_default_registry(device=dev, dtype=torch.float16).create("vit_b_16")
for n in (32, 64, 128):
    host = np.random.default_rng(0).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    tiles = torch.from_numpy(host).to(dev)
    out = torch.empty((n, 768), dtype=torch.float32, device=dev)
    for _ in range(3): ex.forward_device(tiles, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): ex.forward_device(tiles, out)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        ex.forward_device(tiles, out); torch.cuda.synchronize()
    eager_sync = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ex.forward_device(tiles, out)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay(); torch.cuda.synchronize()
    graph_sync = (time.perf_counter() - t0) / 50
    patches = [host[i] for i in range(n)]
    ex.extract_batch(patches)
    t0 = time.perf_counter()
    for _ in range(20): ex.extract_batch(patches, batch_size=32)
    eb = (time.perf_counter() - t0) / 20
    pin = torch.from_numpy(host).pin_memory()
    t0 = time.perf_counter()
    for _ in range(20):
        tiles.copy_(pin, non_blocking=True); g.replay(); o = out.cpu()
    piped = (time.perf_counter() - t0) / 20
    print(f"n={n}: forward back-to-back {eager*1e3:.2f} ms, forward+sync {eager_sync*1e3:.2f} ms, graph replay+sync {graph_sync*1e3:.2f} ms, "
          f"extract_batch {eb*1e3:.2f} ms, pinned H2D + graph + D2H {piped*1e3:.2f} ms")
