#!/usr/bin/env python3
"""Times the fused-LayerNorm GEMM epilogues against the plain store epilogue on the ViT-B shapes (M = 2048 tiles).
Back-to-back launches of one shape: the package sits at its power cap, so these rates are lower than the same kernel reaches inside a forward."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); sp = _lib.current_stream_ptr
M = int(sys.argv[1]) if len(sys.argv) > 1 else 403456
g = torch.Generator().manual_seed(0)
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
out = {}
for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072), ("qkv", 2304, 768), ("fc1", 3072, 768)):
    A = torch.randn(M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).half().to(dev)
    bias = torch.randn(N, generator=g).to(dev); cs = W.float().sum(-1).contiguous()
    O = torch.zeros((M, N), dtype=torch.float16, device=dev)
    rs = torch.ones((M, 2), device=dev); part = torch.empty((M, N // 64, 2), device=dev)
    flop = 2.0 * M * N * K
    plain_epi = 1 if name == "fc1" else 0
    ms_plain = t(lambda: _lib.check(lib.ap_gemm(1, plain_epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, O.data_ptr(), N, 256, 0, sp()), "g"))
    if name in ("proj", "fc2"):
        ms_f = t(lambda: _lib.check(lib.ap_gemm_fused(1, 6, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), None, None, part.data_ptr(), O.data_ptr(), N, 0, sp()), "f"))
    else:
        ms_f = t(lambda: _lib.check(lib.ap_gemm_fused(1, 5 if name == "fc1" else 4, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), cs.data_ptr(), rs.data_ptr(), None, O.data_ptr(), N, 0, sp()), "f"))
    out[name] = {"plain_ms": round(ms_plain, 4), "fused_ms": round(ms_f, 4), "plain_TF": round(flop / ms_plain / 1e9, 1), "fused_TF": round(flop / ms_f / 1e9, 1)}
    del A, W, O, part
print(json.dumps({"M": M, "env": {k: v for k, v in os.environ.items() if k.startswith("AP_")}, **out}))
