#!/usr/bin/env python3
"""K-tile-pair timeline inside one tile of the persistent GEMM (A/B twin, impl 257): is the time lost at the seam or evenly?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
M = 2048 * 197
for name, N, K, epi in (("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("fc1-store", 3072, 768, 0), ("qkv-gelu", 2304, 768, 1)):
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).half()
    W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).half()
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    out = torch.zeros((M, N), device=dev, dtype=torch.float16)
    T = 48
    buf = torch.zeros((2, 256, T, 8), dtype=torch.int64, device=dev)
    for it in range(12):
        if it == 11:
            lib.ap_gemm_trace(buf.data_ptr(), T)
        _lib.check(lib.ap_gemm(1, epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, out.data_ptr(), N, 257, 0, stream))
    torch.cuda.synchronize()
    lib.ap_gemm_trace(None, 0)
    t = buf.cpu().numpy().astype(np.float64) * 0.01
    coarse, fine = t[0], t[1]
    npair = K // 128
    rows = []
    for w in range(256):
        n = int((coarse[w, :, 0] > 0).sum())
        for ti in range(2, n - 1):
            st = fine[w, ti, :npair]
            seg = list(np.diff(st)) + [coarse[w, ti, 1] - st[-1]]
            rows.append(seg + [coarse[w, ti, 4] - coarse[w, ti, 1], (coarse[w, ti, 6] - coarse[w, ti, 5]) / 0.01 / (coarse[w, ti, 1] - coarse[w, ti, 0])])
    r = np.array(rows)
    print(f"== {name} N={N} K={K} epi={epi}: per K-tile PAIR us (mean over {len(r)} tiles): " + " ".join(f"{v:5.2f}" for v in r[:, :npair].mean(0)) +
          f" | epilogue+drain {r[:, npair].mean():.2f} | clock {r[:, npair + 1].mean():.0f} MHz", flush=True)
