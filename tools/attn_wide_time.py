#!/usr/bin/env python3
"""ap_attention at the long-sequence / wide-head shapes of the zoo (vit_h_14: 1370 tokens, 16 heads stored 96 wide; DINOv3 ViT-7B:
201 tokens, 32 heads of 128; Virchow: 257 tokens x 96; conch_v1: 785 x 64): accuracy vs torch on a slice and median launch time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
for (n, T, H, hd) in ((128, 1370, 16, 96), (128, 1370, 16, 128), (512, 257, 16, 96), (256, 201, 32, 128), (256, 785, 12, 64)):
    qkv = (torch.randn((n * T, 3 * H * hd), device=dev, generator=g)).half()
    out = torch.empty((n * T, H * hd), device=dev, dtype=torch.float16)
    for _ in range(3):
        _lib.check(lib.ap_attention(1, qkv.data_ptr(), out.data_ptr(), n, T, H, hd, stream))
    torch.cuda.synchronize()
    q, k, v = qkv[:T].float().view(T, 3, H, hd).permute(1, 2, 0, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, -1) @ v).permute(1, 0, 2).reshape(T, H * hd)
    err = (out[:T].float() - ref).abs().max().item()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            lib.ap_attention(1, qkv.data_ptr(), out.data_ptr(), n, T, H, hd, stream)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    ms = sorted(ts)[3]
    print(f"n={n} T={T} H={H} hd={hd}: {ms:.4f} ms  {4 * T * T * hd * H * n / ms / 1e9:.0f} TF/s  max_abs_err(first image)={err:.2e}", flush=True)
