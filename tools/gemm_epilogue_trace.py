#!/usr/bin/env python3
"""Per-tile timeline of the persistent GEMM's FUSED-LayerNorm epilogues (twin library, impl 257): main loop / drain that opens the
epilogue / epilogue body / gap to the next tile, ViT-B shapes.  `make -C atlaspatch_amd/csrc twin`, then
ATLASPATCH_HIP_LIB=atlaspatch_amd/libatlaspatch_hip_twin.so python tools/gemm_epilogue_trace.py [images]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048) * 197
CASES = (("qkv NORM_STORE", 2304, 768, 4), ("fc1 NORM_GELU", 3072, 768, 5), ("proj RESID_STATS", 768, 768, 6), ("fc2 RESID_STATS", 768, 3072, 6))
for name, N, K, epi in CASES:
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).half()
    W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).half()
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    colsum = W.float().sum(1).contiguous()
    xf = A.float()
    mean, var = xf.mean(1, keepdim=True), xf.var(1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + 1e-6)
    rowstats = torch.cat([rstd, -mean * rstd], 1).contiguous()
    partial = torch.zeros((M, N // 64, 2), device=dev) if epi == 6 else None
    out = torch.zeros((M, N), device=dev, dtype=torch.float16)
    T = 80
    buf = torch.zeros((256, T, 8), dtype=torch.int64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(40):
        if it == 39:
            lib.ap_gemm_trace(buf.data_ptr(), T)
        if it == 38:
            ev[0].record()
        _lib.check(lib.ap_gemm_fused(1, epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), colsum.data_ptr(), rowstats.data_ptr(),
                                     partial.data_ptr() if partial is not None else None, out.data_ptr(), N, 257, stream))
        if it == 38:
            ev[1].record()
    torch.cuda.synchronize()
    lib.ap_gemm_trace(None, 0)
    t = buf.cpu().numpy().astype(np.float64) * 0.01      # us
    ntile = (t[:, :, 0] > 0).sum(1)
    cat = lambda f: np.concatenate([f(w, ntile[w]) for w in range(256) if ntile[w] > 1])
    main = cat(lambda w, n: t[w, 1:n, 1] - t[w, 1:n, 0])
    drain = cat(lambda w, n: t[w, :n, 2] - t[w, :n, 1])
    body = cat(lambda w, n: t[w, :n, 4] - t[w, :n, 2])
    gap = cat(lambda w, n: t[w, 1:n, 0] - t[w, :n - 1, 4])
    clk = cat(lambda w, n: (t[w, 1:n, 6] - t[w, 1:n, 5]) / 0.01 / np.maximum(t[w, 1:n, 1] - t[w, 1:n, 0], 1e-9))
    print(f"{name:18s} N={N} K={K}: {ev[0].elapsed_time(ev[1]):.3f} ms untraced | per tile: main loop {main.mean():6.2f} us ({main.mean() / (K / 64):.3f} / K-tile)  "
          f"drain {drain.mean():5.2f} (p90 {np.percentile(drain, 90):5.2f})  body {body.mean():5.2f}  restart {gap.mean():5.2f}  | shader clock {clk.mean():.0f} MHz", flush=True)
