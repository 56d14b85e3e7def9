import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
os.environ.setdefault("ATLASPATCH_RANDOM_INIT", "0")
from atlaspatch_amd.encoders import build_default_registry
dev = torch.device("cuda:0")
ex = build_default_registry(device=dev, dtype=torch.float16).create("vit_b_16")
n = 2048
tiles = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).to(dev)
out = torch.empty((n, 768), dtype=torch.float32, device=dev)
for _ in range(3): ex.forward_device(tiles, out)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ex.forward_device(tiles, out)
g.replay(); torch.cuda.synchronize()
res = {"eager": [], "graph": []}
for rep in range(5):
    for mode in ("eager", "graph"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            if mode == "eager": ex.forward_device(tiles, out)
            else: g.replay()
        e1.record(); torch.cuda.synchronize()
        res[mode].append(e0.elapsed_time(e1) / 8)
for m in res: print(m, f"{sorted(res[m])[2]:.3f} ms", [round(x, 2) for x in res[m]])
