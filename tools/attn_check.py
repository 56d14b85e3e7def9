#!/usr/bin/env python3
"""ap_attention vs torch fp32 softmax(q k^T / sqrt(d)) v, and its throughput (n images x heads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
dev = torch.device("cuda:0"); lib = _lib.load(); stream = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
def ref(qkv, n, T, H):
    q, k, v = qkv.float().view(n, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(n * T, H * 64)
# f16 / bf16 tolerance: |out| <= 6 here, half an ulp of the output plus the rounding of P to the operand type
for dt, tol in ((torch.float16, 4e-3), (torch.bfloat16, 3e-2), (torch.float32, 2e-5)):
    for (n, T, H) in [c for c in ((3, 197, 12), (2, 50, 16), (1, 257, 12), (5, 1, 12), (2, 785, 12), (1, 1025, 4)) if dt != torch.float32 or c[1] <= 288]:
        qkv = (torch.randn((n * T, 3 * H * 64), device=dev, generator=g) * 1.5).to(dt)
        out = torch.full((n * T, H * 64), float("nan"), device=dev, dtype=dt)
        _lib.check(lib.ap_attention(_lib.torch_dtype_code(dt), qkv.data_ptr(), out.data_ptr(), n, T, H, 64, stream))
        torch.cuda.synchronize()
        err = (out.float() - ref(qkv, n, T, H)).abs().max().item()
        print(f"{'ok  ' if err <= tol else 'FAIL'} {str(dt)[6:]:9s} n={n} T={T} H={H} max_abs_err={err:.3e}", flush=True)
# forced-rescale cases (deferred-max branch): keys whose scores jump far above the running max late in the sequence
for dt, tol in ((torch.float16, 4e-3), (torch.bfloat16, 3e-2)):
    for (n, T, H, spike_at) in ((2, 197, 12, 150), (1, 785, 12, 700), (1, 300, 4, 70)):
        qkv = (torch.randn((n * T, 3 * H * 64), device=dev, generator=g)).to(dt)
        v = qkv.view(n, T, 3, H, 64)
        v[:, spike_at, 1] = v[:, 5, 0] * 6.0          # key `spike_at` aligned with query 5: a huge late score
        v[:, spike_at + 3, 1] *= 5.0
        out = torch.full((n * T, H * 64), float("nan"), device=dev, dtype=dt)
        _lib.check(lib.ap_attention(_lib.torch_dtype_code(dt), qkv.data_ptr(), out.data_ptr(), n, T, H, 64, stream))
        torch.cuda.synchronize()
        err = (out.float() - ref(qkv, n, T, H)).abs().max().item()
        print(f"{'ok  ' if err <= tol else 'FAIL'} {str(dt)[6:]:9s} spike n={n} T={T} H={H} max_abs_err={err:.3e}", flush=True)
for (n, T, H) in ((1024, 197, 12), (256, 785, 12), (512, 197, 16)):
    qkv = (torch.randn((n * T, 3 * H * 64), device=dev, generator=g)).half()
    out = torch.empty((n * T, H * 64), device=dev, dtype=torch.float16)
    for _ in range(3):
        lib.ap_attention(1, qkv.data_ptr(), out.data_ptr(), n, T, H, 64, stream)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.ap_attention(1, qkv.data_ptr(), out.data_ptr(), n, T, H, 64, stream)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = sorted(ts)[2]
    print(f"AP_ATTN_IMPL={os.environ.get('AP_ATTN_IMPL', 'auto')}: f16 n={n} T={T} H={H}: {ms:.4f} ms  ({4 * T * T * 64 * H * n / ms / 1e9:.1f} TF/s, {4 * n * T * H * 64 * 2 / ms / 1e6:.0f} GB/s)")
