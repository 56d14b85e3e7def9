#!/usr/bin/env python3
"""Tile-walk sweep of the persistent GEMM: column-group width (GemmArgs.walk_cols through the variant bits) on the ViT-B shapes
with the fused epilogues, interleaved, median ms per launch.  0 = the kernel's default (whole rows up to 9 column tiles, groups of 6
above)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atlaspatch_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); st = lambda: _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(0)
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048) * 197
for name, N, K, epi, sweep in (("qkv", 2304, 768, 4, (0, 3, 5)), ("fc1", 3072, 768, 5, (0, 3, 4, 12)), ("fc1 f16 wide", 4096, 1024, 5, (0, 4, 8))):
    A = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).half()
    W = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * (2.0 / K ** 0.5)).half()
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    cs = W.float().sum(-1).contiguous()
    rs = torch.stack([torch.rand(M, device=dev, generator=g) + 0.5, torch.rand(M, device=dev, generator=g) - 0.5], -1).contiguous()
    out = torch.empty((M, N), device=dev, dtype=torch.float16)
    def launch(cols):
        impl = 256 | ((cols << 16) << 12)
        if impl >= 2 ** 31: impl -= 2 ** 32
        _lib.check(lib.ap_gemm_fused(1, epi, A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), cs.data_ptr(), rs.data_ptr(), None,
                                     out.data_ptr(), N, impl, st()), "ap_gemm_fused")
    ref = None
    for c in sweep:
        launch(c); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        else: assert torch.equal(ref, out), (name, c)
    t = {c: [] for c in sweep}
    for _ in range(7):
        for c in sweep:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): launch(c)
            e1.record(); torch.cuda.synchronize()
            t[c].append(e0.elapsed_time(e1) / 5)
    print(name, "  ".join(f"cols {c}: {sorted(v)[3]:.4f} ms" for c, v in t.items()), flush=True)
