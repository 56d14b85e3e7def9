#!/usr/bin/env python3
"""A few ap_attention launches of one shape (for rocprofv3 passes): python tools/attn_shape_run.py n T H [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
n, T, H = (int(a) for a in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn((n * T, 3 * H * 64), device=dev, generator=g).half()
out = torch.empty((n * T, H * 64), device=dev, dtype=torch.float16)
for _ in range(reps):
    _lib.check(lib.ap_attention(1, qkv.data_ptr(), out.data_ptr(), n, T, H, 64, stream))
torch.cuda.synchronize()
