#!/usr/bin/env python3
"""The four ViT GEMMs of a SMALL device batch (default 32 tiles: M = 6 304) on every implementation of ap_gemm_fused:
256 = persistent 256 x 256 tiles, 128 = 128 x 128 double buffer, 0 = the dispatcher's choice.
Prints us per launch (hipGraph of `reps` launches, so launch gaps do not count) and checks that all give the same bits.

    python tools/small_gemm_ab.py [tiles] [dim]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib

dev = torch.device("cuda:0"); lib = _lib.load()
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 32
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
M = tiles * 197
g = torch.Generator(device=dev).manual_seed(0)
CASES = (("qkv  NORM_STORE ", 3 * D, D, 4), ("proj RESID_STATS", D, D, 6), ("fc1  NORM_GELU  ", 4 * D, D, 5), ("fc2  RESID_STATS", D, 4 * D, 6))
IMPLS = (256, 128, 0)
total = {i: 0.0 for i in IMPLS}
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
    base = ((torch.rand((M, N), device=dev, generator=g) * 2 - 1)).half()
    outs, times = {}, {}
    for impl in IMPLS:
        out = base.clone()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        sp = s.cuda_stream

        def launch():
            _lib.check(lib.ap_gemm_fused(1, epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), colsum.data_ptr(), rowstats.data_ptr(),
                                         partial.data_ptr() if partial is not None else None, out.data_ptr(), N, impl, sp))
        with torch.cuda.stream(s):
            launch()
            s.synchronize()
            outs[impl] = out.clone()
            reps = 50
            if epi == 6:
                out.copy_(base)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(reps):
                    launch()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s); gr.replay(); e1.record(s); s.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        times[impl] = best
        total[impl] += best
    same = all(torch.equal(outs[IMPLS[0]], outs[i]) for i in IMPLS[1:])
    tf = 2.0 * M * N * K / 1e6
    print(f"{name} M={M} N={N} K={K}: " + "  ".join(f"impl {i}: {times[i]:6.1f} us ({tf / times[i]:5.0f} TF/s)" for i in IMPLS) + f"  bit-identical: {same}", flush=True)
print("per layer: " + "  ".join(f"impl {i}: {total[i]:6.1f} us" for i in IMPLS))
