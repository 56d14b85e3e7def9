#!/usr/bin/env python3
"""ap_attention time on the token counts the encoders have: python tools/attn_time.py [n:T:H[:hd] ...]
(AP_ATTN_NW=5..8 forces the waves per workgroup of the 64-wide kernel; read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
shapes = sys.argv[1:] or ["2048:197:12", "512:257:16", "512:261:12", "512:265:24", "256:785:12", "128:1370:16:96"]
for sh in shapes:
    f = [int(v) for v in sh.split(":")]
    n, T, H = f[:3]; hd = f[3] if len(f) > 3 else 64
    qkv = torch.randn((n * T, 3 * H * hd), device=dev, generator=g).half()
    out = torch.empty((n * T, H * hd), device=dev, dtype=torch.float16)
    for _ in range(3):
        _lib.check(lib.ap_attention(1, qkv.data_ptr(), out.data_ptr(), n, T, H, hd, stream))
    torch.cuda.synchronize()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.ap_attention(1, qkv.data_ptr(), out.data_ptr(), n, T, H, hd, stream)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = sorted(ts)[3]
    print(f"ATTN NW={os.environ.get('AP_ATTN_NW', 'auto')} n={n} T={T} H={H} hd={hd}: {ms:.4f} ms  {4 * T * T * hd * H * n / ms / 1e9:.1f} TF/s  {4 * n * T * H * hd * 2 / ms / 1e6:.0f} GB/s", flush=True)
