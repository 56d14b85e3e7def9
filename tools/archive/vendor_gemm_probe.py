#!/usr/bin/env python3
"""Calibration only (not a product path): what does the vendor library behind torch.matmul reach on the ViT-B/16
GEMM shapes on this box (plain GEMM, no bias / GELU epilogue), next to ap_gemm on the same operands?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
M = 2048 * 197
for dt in (torch.float16, torch.bfloat16):
    for name, N, K, epi in (("qkv", 2304, 768, 0), ("proj", 768, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 0)):
        A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).to(dt)
        W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).to(dt)
        bias = torch.rand(N, device=dev, generator=g) - 0.5
        out = torch.empty((M, N), device=dev, dtype=dt)
        code = _lib.torch_dtype_code(dt)
        def ours():
            _lib.check(lib.ap_gemm(code, epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, out.data_ptr(), N, 0, 0, stream))
        def vendor():
            torch.matmul(A, W.t(), out=out)
        res = {}
        for rnd in range(5):
            for label, fn in (("ap_gemm (bias%s fused)" % ("+GELU" if epi else ""), ours), ("torch.matmul (plain)", vendor)):
                for _ in range(2): fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): fn()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(label, []).append(e0.elapsed_time(e1) / 5)
        print(str(dt)[6:], name, {k: f"{sorted(v)[2]:.3f} ms = {2.0 * M * N * K / sorted(v)[2] / 1e9:.0f} TF" for k, v in res.items()}, flush=True)
