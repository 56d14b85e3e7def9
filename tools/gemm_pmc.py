#!/usr/bin/env python3
"""Runs each ViT-B/16 GEMM shape a few times through ap_gemm (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
M = 1024 * 197
impl = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name, N, K, epi in (("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 0), ("proj", 768, 768, 0)):
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).half()
    W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).half()
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    out = torch.zeros((M, N), device=dev, dtype=torch.float16)
    for it in range(4):
        _lib.check(lib.ap_gemm(1, epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, out.data_ptr(), N, impl, 0, stream))
    torch.cuda.synchronize()
